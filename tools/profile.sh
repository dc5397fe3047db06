#!/bin/bash
# rocprofv3 collection for bench.py on the GPU box.  Writes raw output under gpurun_out/prof/<tag>/
# and a compact summary gpurun_out/prof/<tag>/summary.md (copy that into profiles/).
#   tools/profile.sh <tag> [bench args...]           FRAMES=512 (default: the literal C3 on one GPU) or FRAMES=64 (its 8-GPU shard)
# The traffic record (traffic.json -> profiles/<tag>_traffic.json) carries the build stamp of the library it was taken on
# (bench.build_stamp(): hash of the library's sources + of the loaded .so + git HEAD); bench.py refuses a record of another build.
# Passes are separate on purpose: kernel-trace/stats alone, then one --pmc pass per counter group
# (FETCH_SIZE and WRITE_SIZE do not fit one TCC pass; never mix --pmc with API tracing).
set -u
TAG=${1:-r01}; shift || true
FRAMES=${FRAMES:-512}
ARGS=${@:---steps 20 --warmup 3 --no-cpu --no-extras --frames $FRAMES}
PASSES=${PASSES:-trace fetch write sq1 sq2 tcc pfine}
OUT=$PWD/gpurun_out/prof/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
run() { # name, rocprof flags...
  local name=$1; shift
  case " $PASSES " in *" $name "*) ;; *) return;; esac
  ( cd /tmp && rocprofv3 "$@" --output-format csv -d "$OUT/$name" -- python3 "$OLDPWD/bench.py" $ARGS ) > "$OUT/$name.log" 2>&1
  echo "$name rc=$?" >> "$OUT/passes.log"
}
run trace --kernel-trace --stats
run fetch --kernel-trace --pmc FETCH_SIZE
run write --kernel-trace --pmc WRITE_SIZE
run sq1 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU
run tcc --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
# P_fine: the product kernels at levels = 1 (tools/pfine.py), kernel trace only
case " $PASSES " in *" pfine "*)
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/pfine" -- python3 "$OLDPWD/tools/pfine.py" ) > "$OUT/pfine.log" 2>&1
  echo "pfine rc=$?" >> "$OUT/passes.log";;
esac
# C4 (one 16384^2 frame, level 8) has a script of its own: tools/c4_profile.sh <tag>_c4 -> gpurun_out/prof/<tag>_c4/summary.md
case " $PASSES " in *" c4 "*) tools/c4_profile.sh "${TAG}_c4" > "$OUT/c4.log" 2>&1; echo "c4 rc=$?" >> "$OUT/passes.log";; esac
HGI_PROF_FRAMES=$FRAMES python3 tools/summarize_prof.py "$OUT" > "$OUT/summary.md" 2>"$OUT/summarize.err"
python3 tools/summarize_prof.py "$OUT" --traffic $FRAMES 4096 4 > "$OUT/traffic.json" 2>>"$OUT/summarize.err"
# the raw per-launch counter tables are tens of MiB per pass (gpurun brings back 64 MiB at most): KEEP_RAW=1 keeps them
[ -n "${KEEP_RAW:-}" ] || find "$OUT" -name '*_counter_collection.csv' -delete
cat "$OUT/passes.log"
tail -60 "$OUT/summary.md"
