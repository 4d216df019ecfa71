#!/bin/bash
# Lane utilisation of the step's kernels: SQ_THREAD_CYCLES_VALU against SQ_ACTIVE_INST_VALU (both per launch), normalised by
# a kernel whose lanes are all active (the forward pyramid).
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4_lanes}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ODHIP_PVQ_SERIAL=1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-replay --no-streaming > $OUT/bench_pmc.json 2> $OUT/pmc.err
cd $GRAFT_REPO_ROOT
python - $OUT <<'PY'
import csv, collections, glob, re, sys
o=sys.argv[1]
ctr=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(o+'/pmc/**/t_counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n=re.sub(r'\(.*$','',r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ',''))
        ctr[n][r['Counter_Name']].append(float(r['Counter_Value']))
rows=[]
for n,c in ctr.items():
    if not n.startswith('k_'): continue
    g={k:sum(v)/len(v) for k,v in c.items()}
    if g.get('SQ_ACTIVE_INST_VALU',0) < 1e5: continue
    rows.append((n,g))
base=[g['SQ_THREAD_CYCLES_VALU']/g['SQ_ACTIVE_INST_VALU'] for n,g in rows if n.startswith('k_forward_pyramid64')]
base=base[0] if base else 64.
for n,g in sorted(rows,key=lambda r:-r[1]['SQ_INSTS_VALU']):
    print('%-44s VALU %8.1f M  thread/active %.2f  lanes %.2f' % (n[:44], g['SQ_INSTS_VALU']/1e6, g['SQ_THREAD_CYCLES_VALU']/g['SQ_ACTIVE_INST_VALU'], g['SQ_THREAD_CYCLES_VALU']/g['SQ_ACTIVE_INST_VALU']/base))
PY
