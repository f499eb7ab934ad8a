#!/bin/bash
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_orig.so
for f in scratch/variants/lib_*.so; do
  cp $f $PKG/yolo355/libyolo355.so
  echo "== $f"
  timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_dropin.py -m gpu -q 2>&1 | grep -E "FAILED|passed|failed" | cut -c1-150 | head -20
done
cp /tmp/lib_orig.so $PKG/yolo355/libyolo355.so
