cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_cfg5
timeout 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cfg5 -o p --output-format csv -- python3 tools/bench_configs.py 5op > gpurun_out/prof_cfg5.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/prof_cfg5/**/*kernel_stats.csv', recursive=True)[0]
tot = 0
for r in list(csv.DictReader(open(f)))[:16]:
  print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:7.1f} us  {r['Percentage']}%")
PY
grep cfg5 gpurun_out/prof_cfg5.log
