#!/bin/bash
# sweep the persistent grid size (waves per CU) of the fused kernels; HGI_NO_QUEUE=1 -> one block per tile
for w in ${@:-0 4 8 15 30}; do
  HGI_WAVES_PER_CU=$w python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queue waves/CU=$w', d['value'], d['config']['encode_ms'], d['config']['decode_ms'])"
done
HGI_NO_QUEUE=1 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('static 1 tile/block', d['value'], d['config']['encode_ms'], d['config']['decode_ms'])"
