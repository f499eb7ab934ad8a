#!/bin/bash
# 3-stream throughput with half-size ring tiles (two 4-wave workgroups per CU) for conv5 / conv6 / conv7 (experiment build)
cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod.so
cp scratch/variants/lib_exp.so $PKG/yolo355/libyolo355.so
for round in 1 2 3; do
  Y355_RING_HALF=0 python scratch/layer_times.py full $round thr 2>&1 | grep -v amdgpu.ids
  Y355_RING_HALF=3 python scratch/layer_times.py half67_5 $round thr 2>&1 | grep -v amdgpu.ids
  Y355_RING_HALF=1 python scratch/layer_times.py half67 $round thr 2>&1 | grep -v amdgpu.ids
done
cp /tmp/lib_prod.so $PKG/yolo355/libyolo355.so
python scratch/layer_times.py --summary
