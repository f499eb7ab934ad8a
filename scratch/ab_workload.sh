#!/bin/bash
# GPU box: bench.py --workload $1 with every scratch/variants/lib_*.so and the production library, interleaved twice
cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod.so
cp /tmp/lib_prod.so scratch/variants/lib_prod.so
for round in 1 2; do
for f in scratch/variants/lib_*.so; do
  n=$(basename $f .so); n=${n#lib_}
  cp $f $PKG/yolo355/libyolo355.so
  python bench.py --workload ${1:-slim_fp32} --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$n', d['value'], d['one_stream']['value'], d['roofline']['nms_ms'], d['roofline']['head_ms'])"
done; done
cp /tmp/lib_prod.so $PKG/yolo355/libyolo355.so
