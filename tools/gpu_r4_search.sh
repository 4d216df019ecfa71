#!/bin/bash
# Search-kernel change check: the PVQ test files, then VALU instruction counts and exclusive durations of a step.
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_search}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_pvq_refbands.py tests/test_gpu_pvq_bands.py tests/test_gpu_pipeline.py tests/test_gpu_pvq_ref.py tests/test_gpu_fullsize.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
export ODHIP_PVQ_SERIAL=1
rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-replay --no-streaming > $OUT/bench_pmc.json 2> $OUT/pmc.err
unset ODHIP_PVQ_SERIAL
cd $GRAFT_REPO_ROOT
python - $OUT <<'PY'
import csv, collections, glob, re, sys
o=sys.argv[1]
ctr=collections.defaultdict(list); dur=collections.defaultdict(list)
for f in glob.glob(o+'/pmc/**/t_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=re.sub(r'\(.*$','',r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ',''))
        ctr[n].append(float(r['Counter_Value']))
for f in glob.glob(o+'/pmc/**/t_kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=re.sub(r'\(.*$','',r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ',''))
        dur[n].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
tot=0
for n in sorted(ctr, key=lambda k:-sum(dur[k])):
    if not n.startswith('k_'): continue
    v=sum(ctr[n])/len(ctr[n]); tot+=v
    print('%-44s VALU %8.1f M   %8.1f us' % (n[:44], v/1e6, sum(dur[n])/len(dur[n])/1e3))
print('total VALU per step %.1f M' % (tot/1e6))
PY
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-streaming 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('ms_per_step', round(d['ms_per_step'],3), 'value %.4g' % d['value'], d.get('pipelined_equals_serial'))
for k,v in d['kernels'].items(): print('   ', k, v.get('avg_ms_per_launch'), v.get('exclusive_avg_ms'))"
