# L2 / memory-side counters of the interp (or any) kernel: bash tools/pmc_mem.sh "<profile_run.py args>" tag kernel_substr
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ARGS=$1; TAG=$2; SUB=${3:-interp_}
i=0
for set in "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_WRITE_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  rm -rf gpurun_out/pmcm_${TAG}_$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmcm_${TAG}_$i -o p --output-format csv -- python3 tools/profile_run.py $ARGS --steps 2 > gpurun_out/pmcm_${TAG}_$i.log 2>&1
done
TAG=$TAG SUB=$SUB python3 - <<'PY'
import csv, glob, collections, os
tag, sub = os.environ['TAG'], os.environ['SUB']
for i in range(1,5):
  for f in glob.glob(f'gpurun_out/pmcm_{tag}_{i}/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
      if sub in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(tag, k, f'{sum(v)/len(v):.4g}', len(v))
PY
