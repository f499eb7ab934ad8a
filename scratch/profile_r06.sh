#!/bin/bash
# GPU box: the round 6 rocprofv3 evidence -> gpurun_out/r06_* (copied into profiles/ afterwards).
# kernel stats of bench.py's default run and of --streams 1; configs 3/4; then PMC passes (separate, kernel-trace only).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="--steps 50 --warmup 10 --repeats 5 --no-cpu-baseline --no-sparse --no-other-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_ks3 -- python3 bench.py $B > gpurun_out/r06_bench_under_rocprof.json 2> gpurun_out/r06_ks3.err
cp gpurun_out/r06_ks3/*/*_kernel_stats.csv gpurun_out/r06_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_ks1 -- python3 bench.py $B --streams 1 > gpurun_out/r06_bench_under_rocprof_one_stream.json 2> gpurun_out/r06_ks1.err
cp gpurun_out/r06_ks1/*/*_kernel_stats.csv gpurun_out/r06_kernel_stats_one_stream.csv
for w in slim_fp32 tiny_int8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r06_ks_$w -- python3 bench.py --workload $w --steps 10 --warmup 3 --streams 1 > gpurun_out/r06_bench_$w.json 2> gpurun_out/r06_ks_$w.err
  cp gpurun_out/r06_ks_$w/*/*_kernel_stats.csv gpurun_out/r06_kernel_stats_$w.csv
done
rm -rf gpurun_out/r06_ks3 gpurun_out/r06_ks1 gpurun_out/r06_ks_slim_fp32 gpurun_out/r06_ks_tiny_int8
bash scratch/pmc_traffic.sh > gpurun_out/r06_pmc_traffic.log 2>&1
rm -rf gpurun_out/pmct_FETCH_SIZE gpurun_out/pmct_WRITE_SIZE
bash scratch/pmc_sq.sh r06 > /dev/null 2>&1
rm -rf gpurun_out/pmc_r06 gpurun_out/pmc2_r06
head -12 gpurun_out/r06_kernel_stats_one_stream.csv | cut -c1-150
