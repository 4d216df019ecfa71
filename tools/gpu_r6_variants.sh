#!/bin/bash
# Round 6: same-box A/B of library variants (daala_amd/lib_<name>/libdaalahip.so, python -m daala_amd.build --variant).
# usage: gpu_r6_variants.sh name1 name2 ...   ("default" = daala_amd/lib)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6_overlap
mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-streaming"
run() {
  $B 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
k=d['kernels']
ex=lambda n: (k.get(n) or {}).get('exclusive_avg_ms') or 0
print('%-10s ms_per_step %.3f  pipe==serial %s  pyr_in_step %.3f | exclusive: pyrL %.3f bandsL %.3f invL %.3f pyrC %.3f bandsC %.3f invC %.3f | sum %.3f' % ('$1', d['ms_per_step'], d.get('pipelined_equals_serial'), d['roofline_filter_dct']['in_step']['avg_ms_per_launch'],
  ex('forward_pyramid_luma'), ex('pvq_noref_bands'), ex('dequant_inverse_luma'), ex('forward_pyramid_chroma'), ex('pvq_ref_bands'), ex('dequant_inverse_chroma'), sum(v.get('exclusive_avg_ms') or 0 for v in k.values())))
"
}
for rep in 1 2 3; do
  for v in "$@"; do
    if [ $v = default ]; then unset ODHIP_LIB; else export ODHIP_LIB=$GRAFT_REPO_ROOT/daala_amd/lib_$v/libdaalahip.so; fi
    run $v
  done
done 2>&1 | tee -a $OUT/variants.txt
