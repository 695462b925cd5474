cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for grp in ${W8GROUPS:-1}; do
TUNE=$([ $grp = 1 ] && echo GROUP_ON || echo GROUP_OFF)
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmcg_${grp}_$i -o p --output-format csv -- python3 tools/profile_run.py --steps 2 --tuning $TUNE > gpurun_out/pmcg_${grp}_$i.log 2>&1
done
done
python3 - <<'PY'
import csv, glob, collections
for grp in (0,1):
  for i in range(1,5):
    for f in glob.glob(f'gpurun_out/pmcg_{grp}_{i}/**/*counter_collection.csv', recursive=True):
      acc = collections.defaultdict(list)
      for r in csv.DictReader(open(f)):
        if 'spread_2d_w8' in r['Kernel_Name']:
          acc[r['Counter_Name']].append(float(r['Counter_Value']))
      for k,v in acc.items(): print(grp, k, f'{sum(v)/len(v):.4g}', len(v))
PY
