#!/bin/bash
# build a variant of libyolo355.so into scratch/variants/lib_<name>.so (git-ignored; ships to the GPU box with gpurun)
# usage: build_variant.sh <name> "<EXTRA flags>"
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
name=$1; extra=$2
mkdir -p $ROOT/scratch/variants
make -s -C $ROOT/yolo-compression-and-deployment-in-fpga_amd/csrc -j4 BUILD=/tmp/y355_variant_$name LIB=$ROOT/scratch/variants/lib_$name.so EXTRA="$extra" 2>&1 | grep -E "error|warning|check_kernels" || true
ls -la $ROOT/scratch/variants/lib_$name.so
