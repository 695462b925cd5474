#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -k "3d or config4 or fixed_point or crowded or geometry_sweep or stages or fine_grid or sort_paths or electric or soak" --durations=5 > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
python3 tools/bench_configs.py 4 4s 2>&1 | grep -v amdgpu > $O/configs.txt; cat $O/configs.txt
for t in 1e-2 1e-3; do python3 tools/profile_run.py --type type_1 --grid 256,256,256 --M 3e7 --tol $t --steps 1 > /dev/null; done
python3 - <<'PY' 2>&1 | grep -v amdgpu
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tensorflow-nufft_amd'))
import numpy as np, torch, tensorflow_nufft as tfft
g = torch.Generator(device='cuda').manual_seed(1)
for tol in (1e-1, 1e-2, 1e-3, 1e-4):
  M = 30_000_000
  pts = (torch.rand((M, 3), generator=g, device='cuda') * 2 - 1) * np.pi
  c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
  plan = tfft.Plan('type_1', [256, 256, 256], 'forward', tol=tol)
  for _ in range(2): plan.set_points(pts); plan.execute(c)
  plan.set_timing(True); plan.get_timing()
  for _ in range(3): plan.set_points(pts); plan.execute(c)
  tm = plan.get_timing()
  print(f'3D 256^3 M=3e7 tol {tol:g} w={plan.info().kernel_width}:', ' '.join(f'{k}={v[0]/max(v[1],1)*1e3:.0f}us' for k, v in tm.items() if v[1]))
  plan.close()
PY
bash tools/pmc_kernels.sh cfg4d "--type type_1 --grid 256,256,256 --M 1e8 --tol 1e-4 --one-call" > $O/pmc_cfg4_dense.txt 2>&1
grep -A3 "dense3\|scatter_ranked" $O/pmc_cfg4_dense.txt
cat gpurun_out/full_size_parity.txt
