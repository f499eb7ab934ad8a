#!/bin/bash
# GPU box: interleaved A/B of library variants (scratch/variants/lib_<name>.so) on the headline: r5_ab_libs.sh "<names>" [rounds] [bench args]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod.so
for r in $(seq 1 ${2:-3}); do for v in $1; do
cp scratch/variants/lib_$v.so $PKG/yolo355/libyolo355.so
python bench.py $3 --no-cpu-baseline --no-other-configs --no-sparse --repeats 8 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
km = d['roofline']['kernel_ms']
print('round $r $v: value', d['value'], 'one_stream', d['one_stream']['value'], 'us', {k[:14]: round(1e3 * v, 1) for k, v in km.items()})"
done; done
cp /tmp/lib_prod.so $PKG/yolo355/libyolo355.so
