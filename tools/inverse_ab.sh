timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pipeline.py tests/test_gpu_pvq_bands.py -x -q 2>&1 | tail -2
# (round 5) the A/B switches below exist in the experiments build of the library only
export ODHIP_LIB=${ODHIP_LIB:-$(cd "$(dirname "$0")/.." && pwd)/daala_amd/lib/libdaalahip_exp.so}
for rep in 1 2 3; do
for m in two one; do
if [ $m = one ]; then export ODHIP_INVERSE_X1=1; else unset ODHIP_INVERSE_X1; fi
python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k=d['kernels']
print('$m', round(d['ms_per_step'],3), d.get('pipelined_equals_serial'), ' '.join('%s=%.3f'%(n.split('_')[0][:4]+n.split('_')[-1][:3], v.get('exclusive_avg_ms') or 0) for n,v in k.items()))"
done; done
