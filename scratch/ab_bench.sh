#!/bin/bash
# GPU box: default bench.py (headline only) with every scratch/variants/lib_*.so and the production library, interleaved twice
cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod.so
cp /tmp/lib_prod.so scratch/variants/lib_prod.so
for round in 1 2; do
for f in scratch/variants/lib_*.so; do
  n=$(basename $f .so); n=${n#lib_}
  cp $f $PKG/yolo355/libyolo355.so
  python bench.py --no-cpu-baseline --no-other-configs --no-sparse --repeats 9 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['value'], d['timing']['value_max'], d['one_stream']['value'])"
done; done
cp /tmp/lib_prod.so $PKG/yolo355/libyolo355.so
