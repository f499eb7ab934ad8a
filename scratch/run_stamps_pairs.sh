#!/bin/bash
# GPU box: stamps_ring_pairs.py for every scratch/variants/lib_diag_*.so
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod.so
for f in scratch/variants/lib_diag_*.so; do
  cp $f $PKG/yolo355/libyolo355.so
  for l in ${LAYERS:-7}; do echo "== $f layer $l"; python scratch/stamps_ring_pairs.py $l 2>&1 | grep -v amdgpu.ids | tail -12; done
done
cp /tmp/lib_prod.so $PKG/yolo355/libyolo355.so
