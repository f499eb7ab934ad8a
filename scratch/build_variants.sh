#!/bin/bash
# usage: scratch/build_variants.sh file.hip name1 "-Dflags1" name2 "-Dflags2" ...
# builds scratch/variants/lib_<name>.so = the production library with file.hip recompiled under the flags
set -e
PKG="/root/repo/yolo-compression-and-deployment-in-fpga_amd"
SRC=$1; shift
mkdir -p /root/repo/scratch/variants
make -s -C $PKG/csrc >/dev/null
while [ $# -gt 0 ]; do
  name=$1; flags=$2; shift 2
  ( cd $PKG/csrc; hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off $flags -c $SRC -o /tmp/var_$name.o 2>&1 | grep -E "error" || true;
    objs=""; for f in build/*.o; do b=$(basename $f .o); if [ "$b.hip" == "$SRC" ]; then objs="$objs /tmp/var_$name.o"; else objs="$objs $f"; fi; done
    hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/scratch/variants/lib_$name.so $objs ) &
done
wait
ls -la /root/repo/scratch/variants/
