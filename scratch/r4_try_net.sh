#!/bin/bash
# GPU box: parity of the generic nets on a variant library, then interleaved A/B of scratch/variants/lib_*.so on configs 3 / 4
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod_keep.so
cp scratch/variants/lib_$1.so $PKG/yolo355/libyolo355.so
timeout 1200 python -m pytest tests/test_fp32_models.py tests/test_round2.py -x -q -m gpu 2>&1 | tail -3
cp /tmp/lib_prod_keep.so $PKG/yolo355/libyolo355.so
for w in slim_fp32 tiny_int8; do STREAMS=3 bash scratch/ab_workload.sh $w 2; done
