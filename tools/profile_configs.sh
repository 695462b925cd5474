#!/bin/bash
# rocprofv3 kernel stats of configs 3, 4 and 4-as-type-2 (3 x (set_points + execute) each); run through gpurun.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
run() {  # name, args...
  name=$1; shift
  rm -rf gpurun_out/prof_$name
  timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$name -o p --output-format csv -- python3 tools/profile_run.py "$@" > gpurun_out/prof_$name.log 2>&1
}
run cfg3 --type type_2 --grid 1024,1024 --M 1e7 --tol 1e-6 --steps 3
run cfg4 --type type_1 --grid 256,256,256 --M 1e8 --tol 1e-4 --steps 3
run cfg4t2 --type type_2 --grid 256,256,256 --M 1e8 --tol 1e-4 --steps 3
python3 - <<'PY'
import csv, glob
out = ['# rocprofv3 --kernel-trace --stats -- python3 tools/profile_run.py ... --steps 3   (MI355X, r01)',
       '# kernel | calls | avg_us | total_ms | pct   (torch RNG / fill kernels of the input generation omitted)']
for name, desc in (('cfg3', 'config 3: 2D type-2 1024^2, M=1e7, tol 1e-6, fp32'), ('cfg4', 'config 4: 3D type-1 256^3, M=1e8, tol 1e-4, fp32'),
                   ('cfg4t2', 'config 4 geometry as type 2')):
  f = glob.glob(f'gpurun_out/prof_{name}/**/*kernel_stats.csv', recursive=True)
  out.append(f'## {desc}')
  if not f: out.append('(no stats)'); continue
  for r in csv.DictReader(open(f[0])):
    n = r['Name']
    if 'at::native' in n or 'distribution' in n: continue
    out.append(f"{n[:100]} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.3f} | {r['Percentage']}")
open('gpurun_out/r01_configs_kernel_stats.txt', 'w').write('\n'.join(out) + '\n')
print('\n'.join(out[:40]))
PY
