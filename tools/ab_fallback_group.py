#!/usr/bin/env python3
"""A/B of the cell-grouped fp64-plane fallback of the 3-D w = 7, 8 plans (options.tuning FBGROUP_OFF = the r04 kernel, one
launch per component): kooshball / Gaussian-blob point sets whose dense subproblems the count-filter bound leaves to the
fp64 planes; one-call transforms, HIP-event spread stage, error against a double-precision tol 1e-9 transform.

    python tools/ab_fallback_group.py [--M 3e7,1e7] [--tols 1e-6,1e-5] [--grid 256]
"""
import argparse, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.environ.get('NUFFT_PKG', os.path.join(ROOT, 'tensorflow-nufft_amd')))
import numpy as np, torch
import tensorflow_nufft as tfft
from tensorflow_nufft._lib import TUNE
ap = argparse.ArgumentParser()
ap.add_argument('--M', default='3e7,1e7'); ap.add_argument('--tols', default='1e-6,1e-5'); ap.add_argument('--grid', type=int, default=256)
args = ap.parse_args()
g = torch.Generator(device='cuda').manual_seed(4)
def radial(n):
  ns = 500; nsp = n // ns
  u = torch.rand(nsp, generator=g, device='cuda') * 2 - 1; ph = torch.rand(nsp, generator=g, device='cuda') * 2 * np.pi
  d = torch.stack([torch.sqrt(1 - u * u) * torch.cos(ph), torch.sqrt(1 - u * u) * torch.sin(ph), u], dim=1)
  s = torch.linspace(-np.pi, np.pi, ns + 1, device='cuda')[:ns]
  p = (d[:, None, :] * s[None, :, None]).reshape(-1, 3)
  return p[torch.randperm(p.shape[0], device='cuda', generator=g)]
def blob(n):   # uniform background + 20 % of the points in a Gaussian blob of sigma = 3 fine cells
  p = (torch.rand((n, 3), generator=g, device='cuda') * 2 - 1) * np.pi
  k = n // 5
  p[:k] = 0.4 + torch.randn((k, 3), generator=g, device='cuda') * (3 * np.pi / args.grid)
  return p[torch.randperm(n, device='cuda', generator=g)]
grid = [args.grid] * 3
for Ms in args.M.split(','):
  M = int(float(Ms))
  for name, pts in (('kooshball', radial(M)), ('blob+uniform', blob(M))):
    m = pts.shape[0]
    c = torch.complex(torch.rand(m, generator=g, device='cuda') - .5, torch.rand(m, generator=g, device='cuda') - .5)
    ref = tfft.nufft(c.to(torch.complex128), pts.double(), grid_shape=grid, transform_type='type_1', tol=1e-9)
    for tol in [float(t) for t in args.tols.split(',')]:
      for vname, tune in (('r04 kernel (FBGROUP_OFF)', TUNE['FBGROUP_OFF']), ('cell-grouped', 0)):
        plan = tfft.Plan('type_1', grid, 'forward', tol=tol, tuning=tune)
        for _ in range(2): out = plan.execute_with_points(pts, c)
        plan.set_timing(True); plan.get_timing()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): out = plan.execute_with_points(pts, c)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        tm = plan.get_timing()
        plan.set_points(pts); b = plan.sub_bounds(); live = b[b != 0]
        err = float(torch.linalg.norm(out.to(torch.complex128) - ref) / torch.linalg.norm(ref))
        print(f'{args.grid}^3 M={m:.2e} {name:13s} tol {tol:g} w={plan.info().kernel_width} {vname:26s}: {dt * 1e3:7.3f} ms, spread {tm["spread"][0] / tm["spread"][1] * 1e3:6.0f} us, '
              f'{live.size} {"stacks" if plan.stacks().size else "subproblems"}, {int((live < 0).sum())} on fp64 planes, err {err:.3e}', flush=True)
        plan.close()
    del ref
