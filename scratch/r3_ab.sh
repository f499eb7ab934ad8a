#!/bin/bash
# GPU box: interleaved A/B of scratch/variants/lib_*.so (per-layer HIP-event intervals, one stream; "thr" adds the 3-stream rate)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod.so
rm -f /tmp/lt_*.json
for round in 1 2 3; do
for f in scratch/variants/lib_*.so; do
  n=$(basename $f .so); n=${n#lib_}
  case $n in *diag*) continue;; esac
  cp $f $PKG/yolo355/libyolo355.so
  python scratch/layer_times.py $n $round ${1:-} 2>&1 | grep -v amdgpu.ids
done; done
python scratch/layer_times.py --summary
cp /tmp/lib_prod.so $PKG/yolo355/libyolo355.so
