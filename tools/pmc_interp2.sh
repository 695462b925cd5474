cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace -d gpurun_out/pmcj -o p --output-format csv -- python3 tools/exp_interp_sorted.py > gpurun_out/pmcj.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/pmcj/**/*counter_collection.csv', recursive=True):
  rows = [r for r in csv.DictReader(open(f)) if 'interp_point' in r['Kernel_Name']]
  by = collections.defaultdict(dict)
  for r in rows: by[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
  for d in sorted(by): print(d, {k: f'{v:.3g}' for k, v in by[d].items()})
PY
