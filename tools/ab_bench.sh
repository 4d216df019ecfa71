#!/bin/bash
# Same-box A/B of the bench step between the current build (A) and daala_amd/lib_ab/libdaalahip.so (B):
# alternating runs, ms per step and the stage exclusive times.
for rep in 1 2 3; do
  for which in A B; do
    if [ $which = B ]; then export ODHIP_LIB=$GRAFT_REPO_ROOT/daala_amd/lib_ab/libdaalahip.so; else unset ODHIP_LIB; fi
    python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k=d['kernels']
print('$which', round(d['ms_per_step'],3), d.get('pipelined_equals_serial'), d['roofline_filter_dct']['frac'], ' '.join('%s=%.3f'%(n.split('_')[0][:4]+n.split('_')[-1][:3], v.get('exclusive_avg_ms') or 0) for n,v in k.items()))
"
  done
done
