#!/bin/bash
# hgi_planes_alloc's line-ups on the literal C3 (512 x 4096^2, three composed planes of 8 chunks), one process per row, on the KNOBS
# build with its trace: `python bench.py --no-extras --no-cpu` under
#   (no knob)                                      the shipped policy
#   HGI_PLANES_TWO_CLASSES=1                       a two-class device emulated: only the two largest groups are lined up (-> per offset, alternating)
#   HGI_PLANES_TWO_CLASSES=1 HGI_PLANES_SIDES_ONLY=1   ... forced onto two sides (grid plane 8 + 0)
#   HGI_PLANES_SEARCH_PLANES=2                     the search stopped two planes' worth of chunks beyond the request (shipped: to the budget)
#   HGI_XCD_MODE=0                                 both directions dealt to the XCDs as contiguous eighths (shipped: the encoder only)
# -> profiles/r04_two_classes.txt, r04_c3_xcd_boxes.txt (last block).  Usage: tools/planes_lineup_sweep.sh [out file]
set -u
cd "$(dirname "$0")/.."
export HGI_LIB_PATH=$PWD/rustyhgi_amd/libhgi_hip_knobs.so HGI_PLANES_TRACE=1
out=${1:-gpurun_out/planes_lineup_sweep.txt}; mkdir -p "$(dirname "$out")"; : > "$out"
run() {
  echo "== ${*:-(no knob)}" >> "$out"
  env "$@" timeout -k 10 200 python bench.py --no-extras --no-cpu > gpurun_out/_lineup.json 2> gpurun_out/_lineup.err || { echo "rc=$?" >> "$out"; tail -5 gpurun_out/_lineup.err >> "$out"; return 1; }
  grep "hgi_planes_alloc" gpurun_out/_lineup.err | grep -v "candidate [0-9]\|check, chunks" >> "$out"
  python - >> "$out" <<'PY'
import json
d = json.loads(open("gpurun_out/_lineup.json").read().strip().splitlines()[-1]); c = d["config"]
print("   value %.1f  encode %.4f ms  decode %.4f ms  separated %s  alloc %.2f s" % (d["value"], c["encode_ms"], c["decode_ms"], c["per_rank"]["planes_separated"], c["per_rank"]["planes_alloc_s"][0]))
PY
}
run HGI_LINEUP_SWEEP=shipped && run HGI_PLANES_TWO_CLASSES=1 && run HGI_PLANES_TWO_CLASSES=1 HGI_PLANES_SIDES_ONLY=1 && run HGI_PLANES_SEARCH_PLANES=2 && run HGI_XCD_MODE=0 && run HGI_LINEUP_SWEEP=shipped
cat "$out"
