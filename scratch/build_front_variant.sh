#!/bin/bash
# quick variant of libyolo355.so that differs only in csrc/front.hip (and engine.hip when flags touch it):
#   build_front_variant.sh <name> "<flags>"   -> scratch/variants/lib_<name>.so   (needs an up-to-date csrc/build/)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd); C=$ROOT/yolo-compression-and-deployment-in-fpga_amd/csrc
name=$1; extra=$2
mkdir -p $ROOT/scratch/variants /tmp/fv_$name
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off $extra -c $C/front.hip -o /tmp/fv_$name/front.o 2>/dev/null
objs=""; for f in engine net ops conv3x3 conv3x3_v2 conv3x3_ring convg conv1 comm head_nms; do objs="$objs $C/build/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/scratch/variants/lib_$name.so $objs /tmp/fv_$name/front.o -ldl
ls -la $ROOT/scratch/variants/lib_$name.so | awk '{print $5, $9}'
