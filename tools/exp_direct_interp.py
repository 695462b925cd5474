"""r06: small type-2 calls through tfft.nufft with and without the direct (unsorted) interpolation: us per call."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
from tensorflow_nufft._lib import TUNE
def opts(name):
  o = tfft.Options(); o._internal = {'tuning': TUNE[name]}; return o
def us(call, n=50):
  for _ in range(5): call()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); e0.record()
  for _ in range(n): call()
  e1.record(); e1.synchronize()
  return e0.elapsed_time(e1) / n * 1e3
print(torch.cuda.get_device_name(0))
g = torch.Generator(device='cuda').manual_seed(1)
for dt, rt in ((torch.complex64, torch.float32), (torch.complex128, torch.float64)):
  for grid in ([128, 128], [256, 256], [512, 512], [1024, 1024], [32, 32, 32], [64, 64, 64], [128, 128, 128]):
    for M in (20_000, 100_000, 200_000, 500_000, 1_000_000, 2_000_000):
      rank = len(grid)
      pts = (torch.rand((M, rank), generator=g, device='cuda', dtype=rt) * 2 - 1) * np.pi
      f = torch.complex(torch.rand(grid, generator=g, device='cuda', dtype=rt), torch.rand(grid, generator=g, device='cuda', dtype=rt))
      on, off = opts('DIRECT_ON'), opts('DIRECT_OFF')
      a = us(lambda: tfft.nufft(f, pts, transform_type='type_2', options=on))
      b = us(lambda: tfft.nufft(f, pts, transform_type='type_2', options=off))
      print(f'{"c128" if dt == torch.complex128 else "c64 "} {"x".join(map(str, grid)):>12} M={M:>8}: direct {a:8.1f} us   sorted {b:8.1f} us   {b / a:5.2f}x', flush=True)
  tfft._lib.lib().nufft_hip_op_clear_cache()
