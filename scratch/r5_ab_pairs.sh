#!/bin/bash
# GPU box: interleaved A/B of the settings of Y355_OPT_FUSE_PAIRS on the headline (three handles) and one stream: r5_ab_pairs.sh "<values>" [rounds]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for r in $(seq 1 ${2:-3}); do for v in ${1:-0 1 4}; do
python bench.py --fuse-pairs $v --no-cpu-baseline --no-other-configs --no-sparse --repeats 8 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
km = d['roofline']['kernel_ms']
print('round $r fuse-pairs $v (0 two launches each, 1 conv3 pair fused, 4 both pairs fused, 2 phase schedule): value', d['value'], 'one_stream', d['one_stream']['value'], 'conv3 / conv4 us', [round(1e3 * v, 1) for k, v in km.items() if 'conv3' in k or 'conv4' in k])"
done; done
