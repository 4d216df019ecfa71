#!/bin/bash
# Round 6: the phase of the step in which both chains run filter + DCT kernels (luma inverse beside chroma padding /
# pyramid / preparation): does capping the luma walker's occupancy let the other chain's kernels in?
cd $GRAFT_REPO_ROOT
export ODHIP_LIB=$GRAFT_REPO_ROOT/daala_amd/lib/libdaalahip_exp.so
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-streaming 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['kernels']
print('%-12s ms_per_step %.3f  pipe==serial %s  invL in-step %.3f excl %.3f  bandsC in-step %.3f excl %.3f' % ('$1', d['ms_per_step'], d.get('pipelined_equals_serial'), k['dequant_inverse_luma']['avg_ms_per_launch'], k['dequant_inverse_luma']['exclusive_avg_ms'], k['pvq_ref_bands']['avg_ms_per_launch'], k['pvq_ref_bands']['exclusive_avg_ms']))"; }
for rep in 1 2; do
  run pad0
  for p in 6000 11000 15000 20000 32000; do ODHIP_INV_LDS_PAD=$p run pad$p; done
done
