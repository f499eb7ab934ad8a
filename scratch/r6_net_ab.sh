#!/bin/bash
# GPU box: interleaved A/B of library variants on configs 3 / 4 (bench.py --workload slim_fp32 / tiny_int8): r6_net_ab.sh "<names>" [rounds]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
PKG="yolo-compression-and-deployment-in-fpga_amd"
cp $PKG/yolo355/libyolo355.so /tmp/lib_prod.so
for r in $(seq 1 ${2:-2}); do for w in slim_fp32 tiny_int8; do for v in $1; do
cp scratch/variants/lib_$v.so $PKG/yolo355/libyolo355.so
python bench.py --workload $w --steps 10 --warmup 3 --net-regions 7 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('round $r $w $v: value', d['value'], 'one_stream', d['one_stream']['value'], 'conv frac', d['roofline']['frac'], 'op_ms', [round(1e3 * x, 1) for x in d['roofline']['op_ms']], 'head', d['roofline']['head_ms'], 'nms', d['roofline']['nms_ms'])"
done; done; done
cp /tmp/lib_prod.so $PKG/yolo355/libyolo355.so
