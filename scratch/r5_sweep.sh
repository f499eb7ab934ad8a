#!/bin/bash
# GPU box: headline against --ring-workgroups and --streams with the round-5 (and later) kernels (two rounds, interleaved)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for r in 1 2; do for cfg in "128 3" "96 3" "160 3" "192 3" "128 4" "128 2"; do set -- $cfg
python bench.py --ring-workgroups $1 --streams $2 --no-cpu-baseline --no-other-configs --no-sparse --repeats 8 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('round $r ring-workgroups $1 streams $2: value', d['value'])"
done; done
