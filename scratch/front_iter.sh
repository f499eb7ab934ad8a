#!/bin/bash
# one GPU iteration on the fused front end: parity test, A/B timing (production build), then phase stamps (diagnostic rebuild)
# usage: front_iter.sh "<EXTRA flags of the build under test>"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_front" 2>&1 | tail -3
python scratch/ab_front.py 64 2>&1 | tail -2
python scratch/ab_front.py 64 u8 2>&1 | tail -2
cd yolo-compression-and-deployment-in-fpga_amd/csrc && rm -f build/front.o build/engine.o && make EXTRA="-DFRONT_DIAG=1 $1" 2>&1 | grep -E "error" ; cd ../..
python scratch/stamps_front.py 2>&1 | tail -10
