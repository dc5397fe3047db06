#!/bin/bash
# A/B of the two scheduling modes of the fast kernels under one PMC group
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof/${1:-ab}
mkdir -p $OUT
CTRS=${2:-"SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY"}
( cd /tmp && rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/queue -- python3 $OLDPWD/bench.py --steps 4 --warmup 1 --no-cpu ) > $OUT/queue.log 2>&1
export HGI_NO_QUEUE=1
( cd /tmp && rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $OUT/static -- python3 $OLDPWD/bench.py --steps 4 --warmup 1 --no-cpu ) > $OUT/static.log 2>&1
python3 - <<PY
import csv,glob,collections
for mode in ("queue","static"):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for p in glob.glob("$OUT/%s/**/*counter_collection.csv"%mode, recursive=True):
        for r in csv.DictReader(open(p)):
            k=r["Kernel_Name"]
            if "fast" in k: acc[k.split("(")[-2][-40:] if False else ("enc" if "enc" in k else "dec")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur=collections.defaultdict(list)
    for p in glob.glob("$OUT/%s/**/*kernel_trace.csv"%mode, recursive=True):
        for r in csv.DictReader(open(p)):
            k=r["Kernel_Name"]
            if "fast" in k: dur["enc" if "enc" in k else "dec"].append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
    for k in acc:
        print(mode,k,"us=%.1f"%(sum(dur[k])/len(dur[k])/1e3), {c:"%.3g"%(sum(v)/len(v)) for c,v in sorted(acc[k].items())})
PY
