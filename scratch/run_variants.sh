#!/bin/bash
# on the GPU box: bench every scratch/variants/lib_*.so (one stream, per-layer times)
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_orig.so
for f in scratch/variants/lib_*.so; do
  cp $f $PKG/yolo355/libyolo355.so
  echo "== $f"
  python bench.py --steps 40 --warmup 10 --no-cpu-baseline --streams ${STREAMS:-1} 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('img/s %.0f  frac %.4f  '%(d['value'],r['frac'])+' '.join('%s %.1f'%(k,v['ms']*1000) for k,v in r['layers'].items())+' head %.1f nms %.1f'%(r['head_ms']*1000,r['nms_ms']*1000))"
done
cp /tmp/lib_orig.so $PKG/yolo355/libyolo355.so
