#!/bin/bash
# GPU box: interleaved A/B of the three settings of Y355_OPT_FUSE_PAIRS on the headline (three handles) and one stream
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in 0 1 2; do
python bench.py --fuse-pairs $v --no-cpu-baseline --no-other-configs --no-sparse --repeats 8 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
km = d['roofline']['kernel_ms']
print('round $r fuse-pairs $v (0 two launches, 1 roles, 2 phases): value', d['value'], 'one_stream', d['one_stream']['value'], 'pair/conv3 us', [round(1e3 * v, 1) for k, v in km.items() if 'conv3' in k])"
done; done
