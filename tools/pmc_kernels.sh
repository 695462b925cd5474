#!/bin/bash
# Counter summary of every nufft_hip kernel of one configuration, one rocprofv3 pass per counter set
# (--pmc with --kernel-trace only, as the pool requires). Usage (through gpurun):
#   bash tools/pmc_kernels.sh TAG "<tools/profile_run.py args>" > gpurun_out/TAG.txt
# Traffic: corrected bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: gfx950 tallies 128-byte
# read requests as 64 bytes; both counters are in KiB). SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* are quad-cycles.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=$1; ARGS=$2
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/pmck_${TAG}_$i
  timeout 600 rocprofv3 --pmc $set --kernel-trace -d gpurun_out/pmck_${TAG}_$i -o p --output-format csv -- python3 tools/profile_run.py $ARGS --steps 2 > gpurun_out/pmck_${TAG}_$i.log 2>&1
done
TAG=$TAG ARGS="$ARGS" python3 - <<'PY'
import csv, glob, collections, os, re
tag = os.environ['TAG']
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for i in range(1, 6):
  for f in glob.glob(f'gpurun_out/pmck_{tag}_{i}/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
      n = r['Kernel_Name']
      if 'nufft_hip' not in n: continue
      n = re.sub(r'^void ', '', n).replace('nufft_hip::(anonymous namespace)::', '').replace('nufft_hip::', '')
      acc[n[:110]][r['Counter_Name']].append(float(r['Counter_Value']))
print(f'# rocprofv3 --pmc <set> --kernel-trace -- python3 tools/profile_run.py {os.environ["ARGS"]} --steps 2 (one pass per set), per-dispatch averages')
for k, cs in acc.items():
  m = {c: sum(v) / len(v) for c, v in cs.items()}
  print(f'## {k}   ({max(len(v) for v in cs.values())} dispatches)')
  if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
    print(f"   FETCH_SIZE {m['FETCH_SIZE']:.0f} KiB  WRITE_SIZE {m['WRITE_SIZE']:.0f} KiB  corrected HBM-side traffic (2 FETCH + WRITE) {(2 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024 / 1e6:.1f} MB")
  rest = {c: v for c, v in m.items() if c not in ('FETCH_SIZE', 'WRITE_SIZE')}
  print('   ' + '  '.join(f'{c} {v:.4g}' for c, v in rest.items()))
PY
