#!/bin/bash
# A/B of an environment switch: bench (serial exclusive times, then the overlapped step) with and without it.
cd $GRAFT_REPO_ROOT
for v in "" "$1"; do
  for mode in serial step; do
    if [ $mode = serial ]; then export ODHIP_PVQ_SERIAL=1; else unset ODHIP_PVQ_SERIAL; fi
    env $v timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-streaming --no-replay 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k=d['kernels']
print('[$v] $mode ms', round(d['ms_per_step'],3), 'noref', k['pvq_noref_bands'].get('exclusive_avg_ms'), 'ref', k['pvq_ref_bands'].get('exclusive_avg_ms'))"
  done
done
