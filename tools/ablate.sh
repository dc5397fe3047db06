#!/bin/bash
export HGI_BENCH_NOCHECK=1
for v in ${VARIANTS:-"" _b4}; do
  HGI_LIB_PATH=$PWD/rustyhgi_amd/libhgi_hip$v.so python bench.py --steps 20 --warmup 3 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant[$v]', d['config']['encode_ms'], d['config']['decode_ms'])"
done
