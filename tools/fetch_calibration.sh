#!/bin/bash
# FETCH_SIZE calibration for gathers (tools/ubench/gather_fetch_bench.hip): counter passes + a plain timed run.
# Run through gpurun; prints a summary (copy to profiles/).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
make -C tools/ubench gather_fetch_bench > /dev/null || exit 1
N=${1:-100000000}
O=gpurun_out/fetchcal; rm -rf $O; mkdir -p $O
tools/ubench/gather_fetch_bench $N > $O/timed.txt 2>&1
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o p --output-format csv -- tools/ubench/gather_fetch_bench $N > $O/p$i.log 2>&1
done
N=$N O=$O python3 - <<'PY'
import csv, glob, collections, os
n = int(os.environ['N']); O = os.environ['O']
print(open(f'{O}/timed.txt').read())
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'{O}/p*/**/*counter_collection.csv', recursive=True):
  for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0]
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
print('counter averages per launch (last launch of each kernel is the timed one; both launches are alike):')
for k in ('stream16', 'gather8', 'gather8x2', 'gather16'):
  for kk, d in acc.items():
    if kk.startswith(k + ' ') or kk == k or kk.startswith('void ' + k) or k + '(' in kk or kk.endswith(k):
      line = f'  {k:10s}'
      for c, v in sorted(d.items()):
        line += f' {c}={sum(v)/len(v):.6g}'
      print(line)
      if 'FETCH_SIZE' in d:
        fs = sum(d['FETCH_SIZE']) / len(d['FETCH_SIZE'])   # KiB? rocprofv3 reports FETCH_SIZE in kilobytes
        print(f'             FETCH_SIZE per element: {fs * 1024 / n:.2f} bytes if the unit is KiB, {fs * 1000 / n:.2f} if kB, {fs / n:.4f} if bytes')
PY
