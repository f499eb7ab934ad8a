#!/bin/bash
# GPU box: kernel stats (rocprofv3 --kernel-trace --stats) of one workload, one stream: ks_quick.sh <workload> -> gpurun_out/ks_<workload>.csv
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
w=$1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ksq_$w -- python3 bench.py --workload $w --steps 10 --warmup 3 --streams 1 --net-regions 2 > gpurun_out/ksq_$w.json 2> gpurun_out/ksq_$w.err
cp gpurun_out/ksq_$w/*/*_kernel_stats.csv gpurun_out/ks_$w.csv
rm -rf gpurun_out/ksq_$w
python3 - <<PY
import csv
for r in list(csv.DictReader(open("gpurun_out/ks_$w.csv")))[:16]:
    print("%-86s calls %5s avg %8.1f us" % (r["Name"][:86], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
