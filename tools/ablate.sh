#!/bin/bash
for v in ${VARIANTS:-""}; do
  HGI_LIB_PATH=$PWD/rustyhgi_amd/libhgi_hip$v.so python bench.py --steps 20 --warmup 3 --no-cpu 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant[$v]', d['config']['encode_ms'], d['config']['decode_ms'], d['config']['max_abs_err'])"
done
