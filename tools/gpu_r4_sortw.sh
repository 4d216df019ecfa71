#!/bin/bash
# (round 5) the A/B switches below exist in the experiments build of the library only
export ODHIP_LIB=${ODHIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/daala_amd/lib/libdaalahip_exp.so}
# Sweep of the work-class weights of the with-reference stage (ODHIP_SORT_W = pulses,candidates,searches in quarters).
cd $GRAFT_REPO_ROOT
DEF="4,0,0 4,4,8 4,8,8 4,4,16 4,8,16 4,12,24 2,8,16 4,16,16"
for w in ${WS:-$DEF}; do
  ODHIP_SORT_W=$w ODHIP_PVQ_SERIAL=1 timeout 120 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-streaming --no-replay 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k=d['kernels']
print('$w', 'serial ms', round(d['ms_per_step'],3), 'ref_bands excl', k['pvq_ref_bands'].get('exclusive_avg_ms'), k['pvq_ref_bands'].get('avg_ms_per_launch'))"
done
