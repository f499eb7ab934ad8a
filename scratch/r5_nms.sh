#!/bin/bash
# GPU box, round 5: NMS parity subset on the production library, phase stamps on the -DY355_EXPERIMENTS variant, kernel times
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "nms or head or end_to_end or smoke or dropin or sweep or edge" 2>&1 | tail -3
bash scratch/run_stamps_nms.sh ${1:-nmsx} slim_int8 slim_fp32 tiny_int8
python bench.py --no-cpu-baseline --no-other-configs --no-sparse --repeats 5 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('value', d['value'], 'one_stream', d['one_stream']['value'])
print({k: v for k, v in d['roofline']['kernel_ms'].items() if 'kernel' in k})
print({k: v for k, v in d['roofline']['kernel_ms_in_timed_region'].items() if 'kernel' in k})"
