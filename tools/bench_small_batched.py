"""r06: many small 2-D transforms in one tfft.nufft call (per-item points: dynamic / multi-frame MRI shapes): us per item."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
def us(call, n=20):
  for _ in range(3): call()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize(); e0.record()
  for _ in range(n): call()
  e1.record(); e1.synchronize()
  return e0.elapsed_time(e1) / n * 1e3
print(torch.cuda.get_device_name(0))
g = torch.Generator(device='cuda').manual_seed(1)
for grid in ([128, 128], [256, 256]):
  for M in (10_000, 50_000, 200_000):
    for B in (1, 16, 64, 256):
      pts = (torch.rand((B, M, 2), generator=g, device='cuda') * 2 - 1) * np.pi
      c = torch.complex(torch.rand((B, M), generator=g, device='cuda'), torch.rand((B, M), generator=g, device='cuda'))
      f = torch.complex(torch.rand([B] + grid, generator=g, device='cuda'), torch.rand([B] + grid, generator=g, device='cuda'))
      t1 = us(lambda: tfft.nufft(c, pts, grid_shape=grid, transform_type='type_1'))
      t2 = us(lambda: tfft.nufft(f, pts, transform_type='type_2'))
      print(f'{grid[0]}^2 M={M:>7} B={B:>4}: type 1 {t1:9.1f} us/call = {t1 / B:7.2f} us/item = {B * M / t1 / 1e3:6.2f} Gpts/s | type 2 {t2:9.1f} us/call = {t2 / B:7.2f} us/item = {B * M / t2 / 1e3:6.2f} Gpts/s', flush=True)
  tfft._lib.lib().nufft_hip_op_clear_cache()
