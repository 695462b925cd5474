#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q -k "3d or config4 or fixed_point or crowded or geometry_sweep or fine_grid or tuning_bits or w8_spread or electric" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python3 - <<'PY' 2>&1 | grep -v amdgpu
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tensorflow-nufft_amd'))
import numpy as np, torch, tensorflow_nufft as tfft
g = torch.Generator(device='cuda').manual_seed(4)
M = 100_000_000
pts = (torch.rand((M, 3), generator=g, device='cuda') * 2 - 1) * np.pi
c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
for name, tuning in (('fused (one call)', 0), ('unfused (one call, TUNE_NO_FUSED)', 1)):
  plan = tfft.Plan('type_1', [256, 256, 256], 'forward', tol=1e-4, tuning=tuning)
  for _ in range(2): out = plan.execute_with_points(pts, c)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(5): out = plan.execute_with_points(pts, c)
  torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
  plan.set_timing(True); plan.get_timing()
  for _ in range(3): plan.execute_with_points(pts, c)
  tm = plan.get_timing()
  print(f'cfg4 {name}: {dt*1e3:.3f} ms/step ', ' '.join(f'{k}={v[0]/max(v[1],1)*1e3:.0f}us' for k, v in tm.items() if v[1]))
  plan.close()
del pts, c
for tol in (1e-1, 1e-2, 1e-3, 1e-4, 1e-5):
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
bash tools/pmc_kernels.sh cfg4f "--type type_1 --grid 256,256,256 --M 1e8 --tol 1e-4 --one-call" > $O/pmc_cfg4_fused.txt 2>&1
grep -A3 "dense3\|scatter_ranked" $O/pmc_cfg4_fused.txt
