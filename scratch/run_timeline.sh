#!/bin/bash
# GPU box: kernel trace of the default three-handle run, then scratch/timeline.py on it
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 bench.py --steps 50 --warmup 10 --repeats 3 --no-cpu-baseline --no-sparse --no-other-configs $BENCH_ARGS > gpurun_out/tl_bench.json 2> gpurun_out/tl.err
f=$(ls gpurun_out/tl/*/*_kernel_trace.csv | head -1)
python3 scratch/timeline.py $f 0.05 0.25 | tee gpurun_out/timeline.txt
gzip -c $f > gpurun_out/tl_kernel_trace.csv.gz; rm -rf gpurun_out/tl
