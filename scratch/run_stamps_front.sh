#!/bin/bash
# GPU box: phase stamps of the fused front end for every scratch/variants/lib_*diag.so (built with -DFRONT_DIAG=1)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod.so
for f in scratch/variants/lib_*diag.so; do
  echo "== $f"
  cp $f $PKG/yolo355/libyolo355.so
  python scratch/stamps_front.py $1 2>&1 | grep -v amdgpu.ids | tail -12
done
cp /tmp/lib_prod.so $PKG/yolo355/libyolo355.so
