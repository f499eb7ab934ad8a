#!/bin/bash
# GPU box: headline rate against the number of engine handles and the workgroups per deep-convolution launch
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for st in 2 3 4; do for rw in 0 192 128; do
  python bench.py --streams $st --ring-workgroups $rw --repeats 30 --no-cpu-baseline --no-sparse --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams $st ring_wgs $rw', d['value'], d['roofline']['whole_path_frac'])"
done; done
