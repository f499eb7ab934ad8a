#!/bin/bash
# GPU box: experiment build (scratch/variants/lib_exp.so) with Y355_RING_HALF settings; per-layer times and 3-stream rate
cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod.so
cp scratch/variants/lib_exp.so $PKG/yolo355/libyolo355.so
for round in 1 2; do
for h in 0 4 $((4 + 256*40)) $((4 + 256*100)); do
  Y355_RING_HALF=$h python scratch/layer_times.py half$h $round thr 2>&1 | grep -v amdgpu.ids
done; done
cp /tmp/lib_prod.so $PKG/yolo355/libyolo355.so
python scratch/layer_times.py --summary
