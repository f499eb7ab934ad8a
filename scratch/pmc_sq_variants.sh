#!/bin/bash
# GPU box: scratch/pmc_sq.sh for every scratch/variants/lib_<name>.so named on the command line (kernel filter = $FILT)
cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod_sq.so
for n in "$@"; do
  cp scratch/variants/lib_$n.so $PKG/yolo355/libyolo355.so
  echo "== $n"
  bash scratch/pmc_sq.sh $n "${FILT:-front_kernel}" | tail -3
  cd $GRAFT_REPO_ROOT
done
cp /tmp/lib_prod_sq.so $PKG/yolo355/libyolo355.so
