#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03h; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q -k "3d or config4 or fixed_point or crowded or geometry_sweep or fine_grid or tuning_bits or electric or 32_byte" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python3 - <<'PY' 2>&1 | grep -v amdgpu
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tensorflow-nufft_amd'))
import numpy as np, torch, tensorflow_nufft as tfft
from tensorflow_nufft import _lib
g = torch.Generator(device='cuda').manual_seed(4)
def run(name, M, grid, tol, tuning, one_call=True, steps=5):
  pts = (torch.rand((M, 3), generator=g, device='cuda') * 2 - 1) * np.pi
  c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
  plan = tfft.Plan('type_1', grid, 'forward', tol=tol, tuning=tuning)
  def step():
    if one_call: plan.execute_with_points(pts, c)
    else: plan.set_points(pts); plan.execute(c)
  for _ in range(2): step()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(steps): step()
  torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
  plan.set_timing(True); plan.get_timing()
  for _ in range(3): step()
  tm = plan.get_timing()
  print(f'{name}: {dt*1e3:.3f} ms/step ', ' '.join(f'{k}={v[0]/max(v[1],1)*1e3:.0f}us' for k, v in tm.items() if v[1]), flush=True)
  plan.close()
T = _lib.TUNE
for rep in range(2):
  run('cfg4 M=1e8 grouped (auto), fused', 100_000_000, [256]*3, 1e-4, 0)
  run('cfg4 M=1e8 GROUP_OFF, fused', 100_000_000, [256]*3, 1e-4, T['GROUP_OFF'])
run('cfg4 M=1e8 grouped, two-call (unfused)', 100_000_000, [256]*3, 1e-4, 0, one_call=False)
run('cfg4 M=1e8 GROUP_OFF, two-call (unfused)', 100_000_000, [256]*3, 1e-4, T['GROUP_OFF'], one_call=False)
run('256^3 M=1e7 tol 1e-4 (0.075/cell)', 10_000_000, [256]*3, 1e-4, 0, one_call=False)
run('128^3 M=3e7 tol 1e-4 (1.8/cell) auto', 30_000_000, [128]*3, 1e-4, 0, one_call=False)
run('128^3 M=3e7 tol 1e-4 (1.8/cell) GROUP_OFF', 30_000_000, [128]*3, 1e-4, T['GROUP_OFF'], one_call=False)
run('256^3 M=3e7 tol 1e-5 (w=7)', 30_000_000, [256]*3, 1e-5, 0, one_call=False, steps=3)
run('256^3 M=3e7 tol 1e-2 (w=4) auto', 30_000_000, [256]*3, 1e-2, 0, one_call=False, steps=3)
run('256^3 M=1e8 tol 1e-2 (w=4) auto', 100_000_000, [256]*3, 1e-2, 0, one_call=False, steps=3)
run('256^3 M=1e8 tol 1e-2 (w=4) GROUP_OFF', 100_000_000, [256]*3, 1e-2, T['GROUP_OFF'], one_call=False, steps=3)
PY
