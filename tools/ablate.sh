#!/bin/bash
# A/B timing of experiment builds on ONE box: VARIANTS="- _a _b" ROUNDS=2 tools/ablate.sh
# Each variant is rustyhgi_amd/libhgi_hip<variant>.so (make -C rustyhgi_amd/csrc VARIANT=_a EXTRA='-D...'); "-" is the
# shipped build.  Prints encode / decode ms of the bench step per variant and round (timing-only builds produce wrong
# pixels, hence HGI_BENCH_NOCHECK).
for r in $(seq 1 ${ROUNDS:-2}); do
  for v in ${VARIANTS:--}; do
    [ "$v" = "-" ] && v=""
    HGI_BENCH_NOCHECK=${NOCHECK:-1} HGI_LIB_PATH=$PWD/rustyhgi_amd/libhgi_hip$v.so python bench.py --steps ${STEPS:-30} --warmup 3 --no-cpu 2>gpurun_out/ablate_err.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $r variant[$v] enc %.4f dec %.4f err %d copy %.4f' % (d['config']['encode_ms'], d['config']['decode_ms'], d['config']['max_abs_err'], d['roofline']['copy_same_run']['avg_launch_ms']))"
  done
done
