"""Host enqueue time of plan-level calls (no op layer): single-set and 16-set plans."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
g = torch.Generator(device='cuda').manual_seed(1)
grid = [128, 128]
def host_us(fn, n=300):
  for _ in range(5): fn()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(n): fn()
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  return 1e6 * (t1 - t0) / n
for K, M in ((1, 10000), (16, 10000), (1, 160000)):
  kw = {'num_point_sets': K} if K > 1 else {}
  plan = tfft.Plan('type_1', grid, 'forward', tol=1e-6, **kw)
  pts = (torch.rand((K, M, 2) if K > 1 else (M, 2), generator=g, device='cuda') * 2 - 1) * np.pi
  c = torch.complex(torch.rand((K, M) if K > 1 else (M,), generator=g, device='cuda'), torch.rand((K, M) if K > 1 else (M,), generator=g, device='cuda'))
  out = plan.execute_with_points(pts, c)
  a = host_us(lambda: plan.set_points(pts))
  b = host_us(lambda: plan.execute(c, out=out))
  d = host_us(lambda: plan.execute_with_points(pts, c, out=out))
  print(f'K={K:2d} M={M}: set_points {a:6.1f} us  execute {b:6.1f} us  execute_with_points {d:6.1f} us (host enqueue per call)')
  plan.close()
