"""r06: host-side cost of a tfft.nufft call (enqueue time, GPU left behind) for single and grouped small transforms."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch, ctypes
import tensorflow_nufft as tfft
from tensorflow_nufft import _lib, nufft_ops
g = torch.Generator(device='cuda').manual_seed(1)
grid = [128, 128]
lib = _lib.lib()
orig = lib.nufft_hip_op_compute
acc = {'c': 0.0, 'n': 0}
for B, M in ((1, 10000), (16, 10000), (64, 10000), (1, 160000)):
  pts = (torch.rand((B, M, 2), generator=g, device='cuda') * 2 - 1) * np.pi
  c = torch.complex(torch.rand((B, M), generator=g, device='cuda'), torch.rand((B, M), generator=g, device='cuda'))
  if B == 1: pts, c = pts[0], c[0]
  for _ in range(5): tfft.nufft(c, pts, grid_shape=grid, transform_type='type_1')
  torch.cuda.synchronize()
  n = 200
  t0 = time.perf_counter()
  for _ in range(n): tfft.nufft(c, pts, grid_shape=grid, transform_type='type_1')
  t1 = time.perf_counter()
  torch.cuda.synchronize()
  t2 = time.perf_counter()
  print(f'B={B:3d} M={M:6d}: enqueue {1e6 * (t1 - t0) / n:7.1f} us per call (host), with the GPU drained {1e6 * (t2 - t0) / n:7.1f} us per call', flush=True)
# the C entry alone (ctypes call), same arguments as nufft_ops builds them
import cProfile, pstats
pts = (torch.rand((16, 10000, 2), generator=g, device='cuda') * 2 - 1) * np.pi
c = torch.complex(torch.rand((16, 10000), generator=g, device='cuda'), torch.rand((16, 10000), generator=g, device='cuda'))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): tfft.nufft(c, pts, grid_shape=grid, transform_type='type_1')
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(12)
