#!/bin/bash
# rocprofv3 record of C4 (16384^2 L8 High): tools/c4_profile.sh <tag>  ->  gpurun_out/prof/<tag>/summary.md
# Kernel trace and the two HBM counters in separate passes; the program comes directly after `--`.
set -u
TAG=${1:-r03_c4}
OUT=$PWD/gpurun_out/prof/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
HERE=$PWD
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$HERE/tools/c4_prof.py" ) > "$OUT/trace.log" 2>&1
echo "trace rc=$?" > "$OUT/passes.log"
( cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$HERE/tools/c4_prof.py" ) > "$OUT/fetch.log" 2>&1
echo "fetch rc=$?" >> "$OUT/passes.log"
( cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$HERE/tools/c4_prof.py" ) > "$OUT/write.log" 2>&1
echo "write rc=$?" >> "$OUT/passes.log"
python3 tools/summarize_c4.py "$OUT" > "$OUT/summary.md" 2> "$OUT/summarize.err"
cat "$OUT/passes.log"; grep -h "C4 hipEvents" "$OUT"/*.log; tail -40 "$OUT/summary.md"
