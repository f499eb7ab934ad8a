#!/bin/bash
# GPU box: parity subset on a variant library, then the interleaved A/B of every scratch/variants/lib_*.so
# usage: r4_try.sh <variant-name-to-test> [thr]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod_keep.so
if [ -n "$1" ] && [ "$1" != "-" ]; then
  cp scratch/variants/lib_$1.so $PKG/yolo355/libyolo355.so
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5
  cp /tmp/lib_prod_keep.so $PKG/yolo355/libyolo355.so
fi
bash scratch/r3_ab.sh $2
