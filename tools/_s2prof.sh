cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for t in type_2 type_1; do
  extra=""; [ $t = type_1 ] && extra="--one-call"
  rocprofv3 --kernel-trace --stats -d gpurun_out/s2_$t -o p --output-format csv -- python3 tools/profile_run.py --type $t --grid 256,256,256 --M 1e8 --tol 1e-4 --steps 3 --tuning SORT2_ON $extra > gpurun_out/s2_$t.log 2>&1
  python3 - <<PY
import csv, glob
for f in glob.glob('gpurun_out/s2_$t/**/*kernel_stats.csv', recursive=True):
  rows = list(csv.DictReader(open(f)))
  for r in rows[:14]:
    print('$t', r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1000, 1))
PY
done
