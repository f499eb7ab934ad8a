#!/bin/bash
# GPU box: parity tests (optional: T=1) + one-stream bench line of the in-tree library
if [ "$T" == "1" ]; then timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -2; fi
python bench.py --steps 40 --warmup 10 --no-cpu-baseline --streams ${STREAMS:-1} 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('img/s %.0f  frac %.4f  '%(d['value'],r['frac'])+' '.join('%s %.1f'%(k,v['ms']*1000) for k,v in r['layers'].items())+' head %.1f nms %.1f'%(r['head_ms']*1000,r['nms_ms']*1000))"
