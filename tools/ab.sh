#!/bin/bash
# static (one block per tile) vs persistent queue, N repetitions each
for i in 1 2; do
HGI_NO_QUEUE=1 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('static', d['value'], d['config']['encode_ms'], d['config']['decode_ms'])"
python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('queue ', d['value'], d['config']['encode_ms'], d['config']['decode_ms'])"
done
