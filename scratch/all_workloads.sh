#!/bin/bash
# one line per workload: value and conv roofline fraction
bash scratch/quick.sh
for w in slim_fp32 tiny_bf16 tiny_int8 yolo_v2_bf16 yolo_v3_bf16; do
  python bench.py --workload $w --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('%-14s %9.1f img/s  frac %.4f'%('$w', d['value'], d['roofline']['frac']))"
done
