#!/usr/bin/env python3
"""Wall time per tfft.nufft call of the reference harness's two single-transform 2-D cases (256^2 modes, M = 2e5) and of
config 2 (1024^2, M = 1e7): 5 burn-in calls, 200 timed (config 2: 50), three repetitions. NUFFT_PKG selects a variant build."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.environ.get('NUFFT_PKG', os.path.join(ROOT, 'tensorflow-nufft_amd')))
import numpy as np, torch
import tensorflow_nufft as tfft
g = torch.Generator(device='cuda').manual_seed(3)
for n, M, calls in ((256, 200_000, 200), (1024, 10_000_000, 50)):
  pts = (torch.rand((M, 2), generator=g, device='cuda') * 2 - 1) * np.pi
  c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
  f = torch.complex(torch.rand((n, n), generator=g, device='cuda') - .5, torch.rand((n, n), generator=g, device='cuda') - .5)
  for ttype, src, grid in (('type_1', c, [n, n]), ('type_2', f, None)):
    res = []
    for rep in range(3):
      for _ in range(5):
        tfft.nufft(src, pts, grid_shape=grid, transform_type=ttype)
      torch.cuda.synchronize(); t0 = time.perf_counter()
      for _ in range(calls):
        tfft.nufft(src, pts, grid_shape=grid, transform_type=ttype)
      torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / calls * 1e6)
    print(f'{n}^2 M={M:.0e} {ttype}: ' + ' / '.join(f'{r:.1f}' for r in res) + ' us per call', flush=True)
