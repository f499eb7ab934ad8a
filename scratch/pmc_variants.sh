#!/bin/bash
# on the GPU box: SQ cycle counters of the conv kernels for every scratch/variants/lib_*.so
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_orig.so
for f in scratch/variants/lib_*.so; do
  n=$(basename $f .so)
  cp $f $PKG/yolo355/libyolo355.so
  rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmcv_$n -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --streams 1 > gpurun_out/pmcv_$n.log 2>&1
done
cp /tmp/lib_orig.so $PKG/yolo355/libyolo355.so
python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmcv_*/')):
    fs=glob.glob(d+'*/*_counter_collection.csv')
    if not fs: print(d,'no data'); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k=r['Kernel_Name']
        if 'ring_kernel' not in k: continue
        k=k.split('<')[1].split(', 5, false')[0]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    print('==',d)
    for k,c in acc.items():
        m={n:sum(v)/len(v) for n,v in c.items()}
        print('  %-26s gui/8 %7.0f  wave_cyc/wave %7.0f  wait_any %.2f wait_inst %.2f wait_lds %.3f  lds_active/CU %6.0f conflict %.2f'%(k,m['GRBM_GUI_ACTIVE']/8,m['SQ_WAVE_CYCLES']*4/2048,m['SQ_WAIT_ANY']/m['SQ_WAVE_CYCLES'],m['SQ_WAIT_INST_ANY']/m['SQ_WAVE_CYCLES'],m['SQ_WAIT_INST_LDS']/m['SQ_WAVE_CYCLES'],m['SQ_LDS_IDX_ACTIVE']/256,m['SQ_LDS_BANK_CONFLICT']/max(1,m['SQ_LDS_IDX_ACTIVE'])))
PY
