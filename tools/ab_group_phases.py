#!/usr/bin/env python3
"""How much of spread_2d_w8_group_kernel could overlapping its per-workgroup phases buy? (r03 verdict item 5)
Upper bounds measured with variants that already exist, config 2, one run:
  A  one-call entry (fused records, in-LDS cell sort in every workgroup)              the product
  B  set_points + execute on a fresh plan (strengths gathered, in-LDS cell sort)
  C  execute on records pre-sorted by cell in HBM (no in-LDS sort phase at all)       = B with its sort phase REMOVED,
     which bounds from below what hiding that phase behind other work could reach
and the LDS-pipe model at the clock measured under an LDS-atomic load."""
import ctypes, os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
from tensorflow_nufft import _lib
M = 10_000_000
g = torch.Generator(device='cuda').manual_seed(2)
pts = (torch.rand((M, 2), generator=g, device='cuda') * 2 - 1) * np.pi
c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
mhz = ctypes.c_double(0)
_lib.lib().nufft_hip_debug_shader_clock_mhz(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.byref(mhz))
def spread_us(plan, fn, n=20):
  for _ in range(3): fn()
  plan.set_timing(2); plan.get_timing()
  for _ in range(n): fn()
  t = plan.get_timing()['spread']
  plan.set_timing(False)
  return t[0] / t[1] * 1e3
for rep in range(2):
  plan = tfft.Plan('type_1', [1024, 1024], 'forward', tol=1e-6)
  a = spread_us(plan, lambda: plan.execute_with_points(pts, c))
  plan.close()
  plan = tfft.Plan('type_1', [1024, 1024], 'forward', tol=1e-6, tuning=_lib.TUNE['CELLSORT_OFF'])
  b = spread_us(plan, lambda: (plan.set_points(pts), plan.execute(c)))
  plan.close()
  plan = tfft.Plan('type_1', [1024, 1024], 'forward', tol=1e-6, tuning=_lib.TUNE['CELLSORT_ON'])
  plan.set_points(pts)
  for _ in range(4): plan.execute(c)       # the cell sort runs inside the third execute
  cc = spread_us(plan, lambda: plan.execute(c))
  plan.close()
  model = M / 256 * 13.0 / (mhz.value * 1e6) * 1e6
  print(f'run {rep}: A one-call (fused, in-LDS sort) {a:.1f} us | B two-call (gather, in-LDS sort) {b:.1f} us | C pre-sorted records (gather, no sort phase) {cc:.1f} us'
        f' | LDS-pipe model 13 cycles x {M // 256} points per CU at {mhz.value:.0f} MHz = {model:.1f} us')
