#!/bin/bash
# GPU box: front-end parity tests, then one-stream kernel stats and the default bench line
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_round2.py -m gpu -x -q -k "front or end_to_end or frames or resize or three_handles" > gpurun_out/r3_front_tests.log 2>&1
tail -15 gpurun_out/r3_front_tests.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r3_stats1 -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-sparse --no-other-configs --repeats 5 --streams 1 > gpurun_out/r3_stats1_bench.json 2> gpurun_out/r3_stats1.log
cp gpurun_out/r3_stats1/*/*_kernel_stats.csv gpurun_out/r3_kernel_stats_one_stream.csv
head -16 gpurun_out/r3_kernel_stats_one_stream.csv | cut -c1-160
python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-sparse --no-other-configs --repeats 7 > gpurun_out/r3_bench3.json 2> gpurun_out/r3_bench3.log
python3 -c "
import json
d=json.loads(open('gpurun_out/r3_bench3.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d.get('roofline',{}).get('whole_path_frac'), d.get('layers'))
"
