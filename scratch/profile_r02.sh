#!/bin/bash
# GPU box: the round's profile set -> gpurun_out/r02_* (copy what is to be judged into profiles/)
#   1. rocprofv3 --kernel-trace --stats of the bench command (kernel averages the bench line must agree with)
#   2. HBM traffic per launch (scratch/pmc_traffic.sh: separate FETCH_SIZE / WRITE_SIZE passes)
#   3. SQ / LDS counters per kernel (scratch/pmc_sq.sh)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_stats -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-sparse --no-other-configs --repeats 5 > gpurun_out/r02_stats_bench.json 2> gpurun_out/r02_stats.log
cp gpurun_out/r02_stats/*/*_kernel_stats.csv gpurun_out/r02_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_stats1 -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-sparse --no-other-configs --repeats 5 --streams 1 > gpurun_out/r02_stats1_bench.json 2> gpurun_out/r02_stats1.log
cp gpurun_out/r02_stats1/*/*_kernel_stats.csv gpurun_out/r02_kernel_stats_one_stream.csv
bash scratch/pmc_traffic.sh > gpurun_out/r02_pmc_traffic.log 2>&1
cp gpurun_out/pmc_traffic.json gpurun_out/r02_pmc_traffic.json
bash scratch/pmc_sq.sh r02 > gpurun_out/r02_sq.log 2>&1
head -16 gpurun_out/r02_kernel_stats_one_stream.csv | cut -c1-150
tail -2 gpurun_out/r02_pmc_traffic.log
for w in slim_fp32 tiny_int8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_stats_$w -- python3 bench.py --workload $w --steps 10 --warmup 3 --streams 1 > gpurun_out/r02_bench_$w.json 2> gpurun_out/r02_stats_$w.log
  cp gpurun_out/r02_stats_$w/*/*_kernel_stats.csv gpurun_out/r02_kernel_stats_$w.csv
done
