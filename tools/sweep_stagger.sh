#!/bin/bash
for st in ${@:-0 256}; do for w in 0 30 64 512; do
  HGI_STAGGER=$st HGI_WAVES_PER_CU=$w python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('knob=$st waves=$w', d['value'], d['config']['encode_ms'], d['config']['decode_ms'])"
done; done
