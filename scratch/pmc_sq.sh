#!/bin/bash
# on the GPU box: SQ / LDS counters per kernel of one bench step (production library), rocprofv3 --pmc only with --kernel-trace
# usage: pmc_sq.sh <tag> [kernel-name substring filter]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
tag=${1:-sq}; filt=${2:-}
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-sparse --no-other-configs --streams 1 $BENCH_ARGS > gpurun_out/pmc_$tag.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc2_$tag -- python3 bench.py --steps 3 --warmup 1 --repeats 1 --no-cpu-baseline --no-sparse --no-other-configs --streams 1 $BENCH_ARGS > gpurun_out/pmc2_$tag.log 2>&1
python3 - "$tag" "$filt" <<'PY'
import csv,glob,collections,sys
tag,filt=sys.argv[1],sys.argv[2]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ('gpurun_out/pmc_%s/'%tag,'gpurun_out/pmc2_%s/'%tag):
    for f in glob.glob(d+'*/*_counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name']
            if filt and filt not in k: continue
            acc[k[:70]][r['Counter_Name']].append(float(r['Counter_Value']))
out=open('gpurun_out/pmc_%s_summary.txt'%tag,'w')
for k,c in sorted(acc.items()):
    m={n:sum(v)/len(v) for n,v in c.items()}
    g=lambda n: m.get(n,0.0)
    wc=max(g('SQ_WAVE_CYCLES'),1)
    line='%-70s gui/8 %8.0f busy/32 %8.0f | wave-cycle shares: wait_any %.2f wait_inst %.2f active %.2f | LDS active/CU %7.0f conflict %.2f | insts/wave VALU %6.0f LDS %5.0f SALU %5.0f | mfma_busy/SIMD %7.0f valu_active/wavecyc %.2f lds_active/wavecyc %.2f waves %d' % (
        k,g('GRBM_GUI_ACTIVE')/8,g('SQ_BUSY_CYCLES')/32,g('SQ_WAIT_ANY')/wc,g('SQ_WAIT_INST_ANY')/wc,g('SQ_ACTIVE_INST_ANY')/wc,
        g('SQ_LDS_IDX_ACTIVE')/256,g('SQ_LDS_BANK_CONFLICT')/max(1,g('SQ_LDS_IDX_ACTIVE')),
        g('SQ_INSTS_VALU')/max(1,g('SQ_WAVES')),g('SQ_INSTS_LDS')/max(1,g('SQ_WAVES')),g('SQ_INSTS_SALU')/max(1,g('SQ_WAVES')),
        g('SQ_VALU_MFMA_BUSY_CYCLES')/1024,g('SQ_ACTIVE_INST_VALU')/wc,g('SQ_ACTIVE_INST_LDS')/wc,int(g('SQ_WAVES')))
    print(line); out.write(line+'\n')
PY
