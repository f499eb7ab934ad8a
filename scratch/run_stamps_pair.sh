#!/bin/bash
# GPU box: phase stamps of the fused pair kernel: run_stamps_pair.sh <variant built with -DPAIR_DIAG=1> ["<ring workgroups> <fuse-pairs value>" ...]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod.so
cp scratch/variants/lib_$1.so $PKG/yolo355/libyolo355.so
for g in "${@:2}"; do echo "== workgroups / variant: $g"; python scratch/stamps_pair.py $g 2>&1 | grep -v amdgpu.ids | tail -4; done
cp /tmp/lib_prod.so $PKG/yolo355/libyolo355.so
