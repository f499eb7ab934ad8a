#!/bin/bash
# quick variant of libyolo355.so that differs in ONE csrc file: build_variant_file.sh <name> <file.hip> "<flags>" -> scratch/variants/lib_<name>.so
set -e
ROOT=$(cd $(dirname $0)/.. && pwd); C=$ROOT/yolo-compression-and-deployment-in-fpga_amd/csrc
name=$1; file=$2; extra=$3; stem=$(basename $file .hip)
mkdir -p $ROOT/scratch/variants /tmp/fv_$name
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffp-contract=off $extra -c $C/$file -o /tmp/fv_$name/$stem.o 2>/tmp/fv_$name/err.txt || { tail -3 /tmp/fv_$name/err.txt; exit 1; }
objs=""; for f in engine pipeline net ops conv3x3 conv3x3_ring convg conv1 front frontb comm head_nms peak convpx convr pxpair convpxb; do [ "$f" = "$stem" ] && continue; objs="$objs $C/build/$f.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/scratch/variants/lib_$name.so $objs /tmp/fv_$name/$stem.o -ldl
ls -la $ROOT/scratch/variants/lib_$name.so | awk '{print $5, $9}'
