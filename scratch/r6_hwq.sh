#!/bin/bash
# GPU box: does GPU_MAX_HW_QUEUES (the HIP runtime's hardware queues per device, default 4) lift the ceiling of four handles?
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for r in 1 2; do for cfg in ${SWEEP:-4,128,4 8,128,4 8,128,5 8,128,6 8,96,6 8,128,8 8,64,8}; do IFS=, read q w s <<< "$cfg"
GPU_MAX_HW_QUEUES=$q python bench.py --ring-workgroups $w --streams $s --no-cpu-baseline --no-other-configs --no-sparse --repeats 8 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('round $r hw-queues $q ring-workgroups $w handles $s: value', d['value'], 'min/max', d['timing']['value_min'], d['timing']['value_max'])"
done; done
