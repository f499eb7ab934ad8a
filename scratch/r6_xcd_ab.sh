#!/bin/bash
# GPU box: round 6, VERDICT r5 item 6: the two n-block items of a tile on one XCD (conv5 / conv6 / conv7): wall-clock A/B + FETCH_SIZE of both libs
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
bash scratch/r5_ab_libs.sh "base xcdp" 3 "--steps 20 --warmup 5"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod2.so
for v in base xcdp; do
  cp scratch/variants/lib_$v.so $PKG/yolo355/libyolo355.so
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/xcd_$v -- python3 bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-sparse --no-other-configs --streams 1 > gpurun_out/xcd_$v.log 2>&1
  python3 - $v <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob("gpurun_out/xcd_%s/*/*_counter_collection.csv" % sys.argv[1]):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "ring_kernel" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(sys.argv[1], k, "launches", len(v), "FETCH_SIZE KB mean %.1f -> x2 = %.2f MB" % (sum(v) / len(v), 2 * sum(v) / len(v) / 1024))
PY
done
cp /tmp/lib_prod2.so $PKG/yolo355/libyolo355.so
