#!/bin/bash
# GPU box: parity of the production front end, A/B of scratch/variants/lib_*.so (front kernel = slot 0), stamps of lib_diag.so
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_front or frames" 2>&1 | tail -3
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod.so
rm -f /tmp/lt_*.json
for round in 1 2 3; do
for f in scratch/variants/lib_*.so; do
  n=$(basename $f .so); n=${n#lib_}
  [ "$n" = "diag" ] && continue
  cp $f $PKG/yolo355/libyolo355.so
  python scratch/layer_times.py $n $round ${1:-} 2>&1 | grep -v amdgpu.ids
done; done
python scratch/layer_times.py --summary
if [ -f scratch/variants/lib_diag.so ]; then
  cp scratch/variants/lib_diag.so $PKG/yolo355/libyolo355.so
  python scratch/stamps_front.py 2>&1 | tail -11
fi
cp /tmp/lib_prod.so $PKG/yolo355/libyolo355.so
