#!/bin/bash
# GPU box: experiment build; CUs per conv launch (ring: Y355_RING_GRID, front: Y355_FRONT_CUS, v2: Y355_V2_CUS); per-layer times and 3-stream rate
cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod.so
cp scratch/variants/lib_exp.so $PKG/yolo355/libyolo355.so
for round in 1 2; do
for g in 256 192 128; do
  Y355_RING_GRID=$g python scratch/layer_times.py ring$g $round thr 2>&1 | grep -v amdgpu.ids
  Y355_RING_GRID=$g Y355_FRONT_CUS=$g Y355_V2_CUS=$g python scratch/layer_times.py all$g $round thr 2>&1 | grep -v amdgpu.ids
done; done
cp /tmp/lib_prod.so $PKG/yolo355/libyolo355.so
python scratch/layer_times.py --summary
