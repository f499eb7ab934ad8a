#!/bin/bash
# GPU box: headline rate against the workgroups per deep-convolution launch (three handles), interleaved rounds
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for round in 1 2 3; do for rw in ${RWS:-96 112 128 144 160 192}; do
  python bench.py --streams 3 --ring-workgroups $rw --repeats 30 --no-cpu-baseline --no-sparse --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $round ring_wgs $rw', d['value'], d['roofline']['whole_path_frac'])"
done; done
