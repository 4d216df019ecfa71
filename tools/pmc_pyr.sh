cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_pyr; rm -rf $O; mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $O/g$i -o t -- python $GRAFT_REPO_ROOT/tools/pyr_only.py > $O/g$i.log 2>&1 || tail -3 $O/g$i.log
done
python - <<'PY'
import csv, collections, os, glob
root=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pmc_pyr'
for f in sorted(glob.glob(root+'/g*/t_counter_collection.csv')):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'k_forward_pyramid' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print("%-24s %16.0f  (n=%d)"%(k, sum(v)/len(v), len(v)))
PY
