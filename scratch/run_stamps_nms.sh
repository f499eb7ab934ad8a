#!/bin/bash
# GPU box: NMS phase stamps (library variant built with -DY355_EXPERIMENTS): run_stamps_nms.sh <variant> [workloads...]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod.so
cp scratch/variants/lib_$1.so $PKG/yolo355/libyolo355.so
for w in ${@:2}; do echo "== $w"; python scratch/stamps_pairs.py $w 2>&1 | grep -v amdgpu.ids | tail -5; done
cp /tmp/lib_prod.so $PKG/yolo355/libyolo355.so
