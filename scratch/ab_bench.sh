#!/bin/bash
# GPU box: interleaved A/B of scratch/variants/lib_*.so on bench.py's headline (three handles): ab_bench.sh [rounds] [extra bench args]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod.so
for round in $(seq 1 ${1:-3}); do
for f in scratch/variants/lib_*.so; do
  n=$(basename $f .so); n=${n#lib_}
  cp $f $PKG/yolo355/libyolo355.so
  python bench.py --steps 50 --warmup 10 --repeats 9 --no-cpu-baseline --no-sparse --no-other-configs ${@:2} 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$n round $round: %.0f img/s  whole_path_frac %.4f  one_stream %.0f  conv sum %.1f us' % (d['value'], r['whole_path_frac'], d['one_stream']['value'], 1e3 * sum(list(r['kernel_ms'].values())[:9])))"
done; done
cp /tmp/lib_prod.so $PKG/yolo355/libyolo355.so
