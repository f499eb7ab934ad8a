#!/bin/bash
# GPU box: interleaved A/B of scratch/variants/lib_*.so on another workload of bench.py (one stream): ab_workload.sh <workload> [rounds]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod.so
for round in $(seq 1 ${2:-3}); do
for f in scratch/variants/lib_*.so; do
  n=$(basename $f .so); n=${n#lib_}
  cp $f $PKG/yolo355/libyolo355.so
  python bench.py --workload $1 --steps 10 --warmup 3 --streams ${STREAMS:-1} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$n round $round: %.0f img/s  conv frac %.4f  head_ms %s nms_ms %s' % (d['value'], r['frac'], r.get('head_ms'), r.get('nms_ms')))"
done; done
cp /tmp/lib_prod.so $PKG/yolo355/libyolo355.so
