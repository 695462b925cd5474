#!/bin/bash
# Regenerates every r03 artefact under profiles/ in one gpurun call (results land in gpurun_out/profiles_r03/, to be
# copied into profiles/): the bench line, rocprofv3 kernel stats of the bench command (the per-kernel averages that
# roofline.kernel_ms must agree with), counter summaries of configs 2, 3 and 4 (one --pmc pass per counter set,
# --kernel-trace only), HIP-event stage times of configs 1-5, per-kernel averages of configs 3, 4 and 4-as-type-2,
# the 3-D tolerance sweep. GPU suite first, so that the numbers belong to a green tree.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/profiles_r03; rm -rf $O; mkdir -p $O
rm -f gpurun_out/full_size_parity.txt
timeout 1800 python -m pytest tests -m gpu -x -q --durations=6 > $O/r03_gpu_suite.txt 2>&1; tail -12 $O/r03_gpu_suite.txt
cp gpurun_out/full_size_parity.txt $O/r03_full_size_parity.txt
timeout 900 python3 bench.py > $O/r03_bench.json 2> $O/bench.err; cut -c1-400 $O/r03_bench.json
rm -rf gpurun_out/prof_bench3
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench3 -o runc --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/prof_bench3.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras   (MI355X, r03)"; python3 tools/kstats.py gpurun_out/prof_bench3 14; tail -c 400 gpurun_out/prof_bench3.log | grep -o '"kernel_ms": [0-9.]*' | sed 's/^/# same run, HIP events in bench.py: /'; } > $O/r03_bench_kernel_stats.txt
cat $O/r03_bench_kernel_stats.txt
bash tools/pmc_kernels.sh cfg2 "--type type_1 --grid 1024,1024 --M 1e7 --tol 1e-6 --one-call" > $O/r03_pmc_cfg2.txt 2>&1
bash tools/pmc_kernels.sh cfg3 "--type type_2 --grid 1024,1024 --M 1e7 --tol 1e-6 --one-call" > $O/r03_pmc_cfg3.txt 2>&1
# (config 4 as shipped since the two-level sort: unfused records; r03_pmc_cfg4_fused.txt is the same call with --tuning SORT2_OFF)
bash tools/pmc_kernels.sh cfg4 "--type type_1 --grid 256,256,256 --M 1e8 --tol 1e-4 --one-call" > $O/r03_pmc_cfg4_sort2.txt 2>&1
bash tools/pmc_kernels.sh cfg4f "--type type_1 --grid 256,256,256 --M 1e8 --tol 1e-4 --one-call --tuning SORT2_OFF" > $O/r03_pmc_cfg4_fused.txt 2>&1
grep -A2 "spread_2d_w8_group\|interp_point\|dense3" $O/r03_pmc_cfg2.txt $O/r03_pmc_cfg3.txt $O/r03_pmc_cfg4_sort2.txt $O/r03_pmc_cfg4_fused.txt | cut -c1-260
python3 tools/sort2_ab.py --big 2>&1 | grep -v amdgpu > $O/r03_sort2_ab.txt; tail -8 $O/r03_sort2_ab.txt | cut -c1-200
python3 tools/bench_configs.py 2 3 4 4t2 5 5s 5op 1 3d5 3d6 2>&1 | grep -v amdgpu > $O/r03_configs.txt; cat $O/r03_configs.txt
for cfg in "type_2 1024,1024 1e7 1e-6 cfg3" "type_1 256,256,256 1e8 1e-4 cfg4" "type_2 256,256,256 1e8 1e-4 cfg4t2"; do
  set -- $cfg
  rm -rf gpurun_out/prof3_$5
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof3_$5 -o run --output-format csv -- python3 tools/profile_run.py --type $1 --grid $2 --M $3 --tol $4 --steps 3 --one-call > gpurun_out/prof3_$5.log 2>&1
  echo "== $5: $1 grid $2 M=$3 tol=$4 (rocprofv3 --kernel-trace --stats, 3 x nufft_hip_execute_with_points)" >> $O/r03_configs_kernel_stats.txt
  python3 tools/kstats.py gpurun_out/prof3_$5 10 | grep -v "at::native" >> $O/r03_configs_kernel_stats.txt
done
cat $O/r03_configs_kernel_stats.txt
python3 - <<'PY' 2>&1 | grep -v amdgpu > $O/r03_3d_tol_sweep.txt
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tensorflow-nufft_amd'))
import numpy as np, torch, tensorflow_nufft as tfft
print('# 3-D type 1, 256^3 modes (512^3 fine cells), M = 3e7 uniform points, complex64: HIP-event stage times per call (set_points + execute)')
g = torch.Generator(device='cuda').manual_seed(1)
M = 30_000_000
pts = (torch.rand((M, 3), generator=g, device='cuda') * 2 - 1) * np.pi
c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
for tol in (1e-1, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6):
  plan = tfft.Plan('type_1', [256, 256, 256], 'forward', tol=tol)
  for _ in range(2): plan.set_points(pts); plan.execute(c)
  plan.set_timing(True); plan.get_timing()
  for _ in range(3): plan.set_points(pts); plan.execute(c)
  tm = plan.get_timing()
  i = plan.info()
  print(f'tol {tol:g} w={i.kernel_width} tile={list(i.tile_dims)}:', ' '.join(f'{k}={v[0]/max(v[1],1)*1e3:.0f}us' for k, v in tm.items() if v[1]))
  plan.close()
PY
cat $O/r03_3d_tol_sweep.txt
(cd tools/ubench && ./lds_pattern_bench) > $O/r03_lds_pattern_ubench.txt 2>&1
