#!/bin/bash
# on the GPU box: per-layer HIP-event times (us, median of rounds, one stream) and, with "thr", the 3-stream throughput of
# every scratch/variants/lib_*.so, interleaved rounds in one call; the production library is variant "prod"
cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod.so
cp /tmp/lib_prod.so scratch/variants/lib_prod.so
for round in 1 2 3; do
for f in scratch/variants/lib_*.so; do
  n=$(basename $f .so); n=${n#lib_}
  cp $f $PKG/yolo355/libyolo355.so
  python scratch/layer_times.py $n $round ${1:-} 2>&1 | grep -v amdgpu.ids
done; done
cp /tmp/lib_prod.so $PKG/yolo355/libyolo355.so
python scratch/layer_times.py --summary
