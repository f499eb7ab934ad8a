#!/bin/bash
# GPU box: conv3_2 / conv4_2 on convpx (production) against the ring kernel (Y355_NO_PX_MASK, library built with -DY355_EXPERIMENTS)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod.so
cp scratch/variants/lib_engx.so $PKG/yolo355/libyolo355.so
for round in 1 2; do for m in 0 32 8 40; do
  Y355_NO_PX_MASK=$m python bench.py --steps 50 --warmup 10 --repeats 9 --no-cpu-baseline --no-sparse --no-other-configs 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; k = r['kernel_ms']
print('mask $m round $round: %.0f img/s one stream %.0f | conv3_2 %.1f conv4_2 %.1f us' % (d['value'], d['one_stream']['value'], 1e3 * k['conv3_2'], 1e3 * k['conv4_2']))"
done; done
cp /tmp/lib_prod.so $PKG/yolo355/libyolo355.so
