#!/bin/bash
# GPU box: headline (through the pipeline entry point) against handles and workgroups per launch, round-6 kernels (two rounds, interleaved)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for r in 1 2; do for cfg in ${SWEEP:-128,3 128,4 96,4 160,4 192,4 128,5 96,5 128,6}; do set -- ${cfg/,/ }
python bench.py --ring-workgroups $1 --streams $2 --no-cpu-baseline --no-other-configs --no-sparse --repeats 8 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('round $r ring-workgroups $1 handles $2: value', d['value'], 'min/max', d['timing']['value_min'], d['timing']['value_max'])"
done; done
