#!/bin/bash
# L2 <-> memory-fabric (EA) counters of the two placements: the same bench step with every launch reading
# one HBM region and writing another (--placement planes) and with all three frame stacks in ONE region (--placement
# same-region).  Separate rocprofv3 --pmc passes per counter group; writes gpurun_out/placement_counters.md.
set -u
OUT=$PWD/gpurun_out/prof/placement
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu --no-extras"
for place in planes same-region; do
  for grp in "rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
             "wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum" \
             "tag TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_REQ_sum TCC_BUSY_sum"; do
    set -- $grp; name=$1; shift
    ( cd /tmp && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/${place}_$name" -- python3 "$OLDPWD/bench.py" $ARGS --placement $place ) > "$OUT/${place}_$name.log" 2>&1
    echo "$place $name rc=$?"
  done
done
python3 - "$OUT" <<'PY' > gpurun_out/placement_counters.md
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
rows = {}
for place in ("planes", "same-region"):
    acc = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
    for grp in ("rd", "wr", "tag"):
        for path in glob.glob(os.path.join(out, "%s_%s" % (place, grp), "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                k = r["Kernel_Name"]
                if "k_enc_tiles<1" in k or "k_dec_tiles<1" in k:
                    acc["enc" if "k_enc" in k else "dec"][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for path in glob.glob(os.path.join(out, "%s_%s" % (place, grp), "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                k = r["Kernel_Name"]
                if "k_enc_tiles<1" in k or "k_dec_tiles<1" in k:
                    dur["enc" if "k_enc" in k else "dec"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows[place] = (acc, dur)
print("# EA (L2 <-> fabric) counters per launch: different regions (`planes`) vs one region (`same-region`)\n")
print("64 x 4096^2 L4 Medium, averages per launch over the profiled passes; LEVEL / REQ = mean cycles a request is outstanding.\n")
for kern in ("enc", "dec"):
    print("## k_%s_tiles\n" % kern)
    names = sorted(set(rows["planes"][0][kern]) | set(rows["same-region"][0][kern]))
    print("| counter | different regions | one region | ratio |")
    print("|---|---|---|---|")
    for place in ():
        pass
    d0 = rows["planes"][1][kern]; d1 = rows["same-region"][1][kern]
    if d0 and d1:
        a, b = sum(d0) / len(d0) / 1e3, sum(d1) / len(d1) / 1e3
        print("| launch duration (us, under the profiler) | %.1f | %.1f | %.3f |" % (a, b, b / a))
    vals = {}
    for n in names:
        v0 = rows["planes"][0][kern].get(n); v1 = rows["same-region"][0][kern].get(n)
        if v0 and v1:
            a, b = sum(v0) / len(v0), sum(v1) / len(v1)
            vals[n] = (a, b)
            print("| %s | %.4g | %.4g | %.3f |" % (n, a, b, b / a if a else float("nan")))
    for req, lvl in (("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_LEVEL_sum"), ("TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_LEVEL_sum")):
        if req in vals and lvl in vals:
            a, b = vals[lvl][0] / vals[req][0], vals[lvl][1] / vals[req][1]
            print("| %s / %s (cycles outstanding) | %.1f | %.1f | %.3f |" % (lvl, req, a, b, b / a))
    print()
PY
cat gpurun_out/placement_counters.md
