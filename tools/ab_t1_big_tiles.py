"""A/B at BASELINE config 2 (2-D type 1, 1024^2, M = 1e7, tol 1e-6): 32 x 32 tiles (4096-tile scattered sort +
grouped spread, two workgroups per CU) against 64 x 64 tiles (options.tuning T1_BIG_TILES: one staged scatter
pass into 1024 tiles + the grouped spread on 82 KB planes, one workgroup per CU). Two-call form (set_points +
execute, strengths gathered) for both -- the fused records of the one-call form exist for 32 x 32 only."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
from tensorflow_nufft import _lib
g = torch.Generator(device='cuda').manual_seed(2)
outs = {}
for M, N in ((10_000_000, 1024), (1_000_000, 512)):
  pts = (torch.rand((M, 2), generator=g, device='cuda') * 2 - 1) * np.pi
  c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
  for name, tuning in (('32x32 tiles', 0), ('64x64 tiles', _lib.TUNE['T1_BIG_TILES'])):
    plan = tfft.Plan('type_1', [N, N], 'forward', tol=1e-6, tuning=tuning)
    i = plan.info()
    for _ in range(3):
      plan.set_points(pts); out = plan.execute(c)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
      plan.set_points(pts); plan.execute(c)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    plan.set_timing(True); plan.get_timing()
    for _ in range(10):
      plan.set_points(pts); plan.execute(c)
    tm = plan.get_timing()
    outs[name] = out
    print(f'M={M:.0e} N={N} {name} (tile {list(i.tile_dims)[:2]}, max_sub {i.max_subproblem_size}): {dt*1e3:.3f} ms/step  ',
          ' '.join(f'{k}={v[0]/max(v[1],1)*1e3:.0f}us' for k, v in tm.items() if v[1]), flush=True)
    plan.close()
  d = (outs['32x32 tiles'] - outs['64x64 tiles']).abs().double().pow(2).sum().sqrt() / outs['32x32 tiles'].abs().double().pow(2).sum().sqrt()
  print(f'   rel-l2 between the two: {float(d):.2e}')
