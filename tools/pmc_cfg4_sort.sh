cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc4_$c
  timeout 600 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc4_$c -o p --output-format csv -- python3 tools/profile_run.py --type type_1 --grid 256,256,256 --M 1e8 --tol 1e-4 --steps 2 --one-call > gpurun_out/pmc4_$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
for c in ('FETCH_SIZE','WRITE_SIZE'):
  for f in glob.glob(f'gpurun_out/pmc4_{c}/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
      n = r['Kernel_Name']
      if 'nufft_hip' in n: acc[n.split('(')[0][-60:]].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(c, k, f'{sum(v)/len(v)/1024:.1f} MiB (KiB units)', len(v))
PY
