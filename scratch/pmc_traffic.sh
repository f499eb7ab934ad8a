#!/bin/bash
# GPU box: HBM traffic per launch (separate FETCH_SIZE / WRITE_SIZE passes, kernel-trace only) -> gpurun_out/pmc_traffic.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmct_$c -- python3 bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-sparse --no-other-configs --streams 1 > gpurun_out/pmct_$c.log 2>&1
  cp gpurun_out/pmct_$c/*/*_counter_collection.csv gpurun_out/pmc_${c}_counter_collection.csv
done
python3 - <<'PY'
import csv, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for r in csv.DictReader(open("gpurun_out/pmc_%s_counter_collection.csv" % c)):
        if r["Counter_Name"] == c:
            acc[r["Kernel_Name"]][c].append(float(r["Counter_Value"]))
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only), bench.py --steps 2 --warmup 1 --streams 1, B=64; bytes = (2*FETCH_SIZE + WRITE_SIZE) KB * 1024 (gfx950 FETCH_SIZE counts 128-B requests at 64 B)", "kernels": {}}
for k, v in acc.items():
    f = sum(v["FETCH_SIZE"]) / max(1, len(v["FETCH_SIZE"])); w = sum(v["WRITE_SIZE"]) / max(1, len(v["WRITE_SIZE"]))
    out["kernels"][k] = {"FETCH_SIZE_KB_mean": round(f, 1), "launches": len(v["FETCH_SIZE"]), "WRITE_SIZE_KB_mean": round(w, 1),
                         "hbm_bytes_per_launch": int((2 * f + w) * 1024)}
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
for k, v in out["kernels"].items():
    if "ring_kernel<256, 128" in k or "conv1_fast" in k: print(k[:70], v)
PY
