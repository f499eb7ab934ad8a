#!/bin/bash
# GPU box: phase stamps of variants of the fused pair kernel: run_stamps_pair_abl.sh "<ring workgroups> <fuse-pairs value>" <variant> ...
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod.so
for v in "${@:2}"; do
  cp scratch/variants/lib_$v.so $PKG/yolo355/libyolo355.so
  echo "== $v"; python scratch/stamps_pair.py $1 2>&1 | grep -v amdgpu.ids | tail -3
done
cp /tmp/lib_prod.so $PKG/yolo355/libyolo355.so
