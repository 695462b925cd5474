# PMC counters of the wide spread kernel: bash tools/pmc_wide.sh "<profile_run.py args>" tag
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ARGS=${1:---type type_1 --grid 1024,1024 --M 1e7 --tol 1e-9 --double}
TAG=${2:-w11_2d}
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcw_${TAG}_$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmcw_${TAG}_$i -o p --output-format csv -- python3 tools/profile_run.py $ARGS --steps 2 > gpurun_out/pmcw_${TAG}_$i.log 2>&1
done
TAG=$TAG python3 - <<'PY'
import csv, glob, collections, os
tag = os.environ['TAG']
for i in range(1,5):
  for f in glob.glob(f'gpurun_out/pmcw_{tag}_{i}/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
      if 'spread_wide' in r['Kernel_Name'] or 'interp_' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(tag, k, f'{sum(v)/len(v):.4g}', len(v))
PY
