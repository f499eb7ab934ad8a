#!/bin/bash
cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod.so
cp scratch/variants/lib_exp.so $PKG/yolo355/libyolo355.so
python scratch/nms_stamps.py 2>&1 | grep -v amdgpu.ids
cp /tmp/lib_prod.so $PKG/yolo355/libyolo355.so
