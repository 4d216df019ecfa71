cd $GRAFT_REPO_ROOT
for v in new base new base; do
  if [ $v = base ]; then export ODHIP_LIB=$GRAFT_REPO_ROOT/daala_amd/lib/libdaalahip_base.so; else unset ODHIP_LIB; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-streaming 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernels']
print('$v: ms_per_step %.3f' % d['ms_per_step'], 'noref in-step %.3f excl %.3f' % (k['pvq_noref_bands']['avg_ms_per_launch'], k['pvq_noref_bands']['exclusive_avg_ms']), 'pipe==serial', d['pipelined_equals_serial'])
"
done
