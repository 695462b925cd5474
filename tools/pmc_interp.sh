# SQ counters of the interp kernel (config 3); run through gpurun.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmci_$i -o p --output-format csv -- python3 tools/profile_run.py --type type_2 --steps 2 > gpurun_out/pmci_$i.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for i in range(1,5):
  for f in glob.glob(f'gpurun_out/pmci_{i}/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
      if 'interp_point' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(k, f'{sum(v)/len(v):.4g}', len(v))
PY
