"""Small driver for rocprofv3: K x (set_points + execute) of one config."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
import argparse
ap = argparse.ArgumentParser()
ap.add_argument('--type', default='type_1'); ap.add_argument('--grid', default='1024,1024')
ap.add_argument('--M', type=float, default=1e7); ap.add_argument('--tol', type=float, default=1e-6)
ap.add_argument('--method', type=int, default=0); ap.add_argument('--S', type=int, default=0)
ap.add_argument('--steps', type=int, default=5); ap.add_argument('--ntransf', type=int, default=1)
ap.add_argument('--acc', type=int, default=0)
ap.add_argument('--tuning', default='', help='comma-separated nufft_hip_options.tuning bits, e.g. GROUP_OFF,ROCFFT')
ap.add_argument('--double', action='store_true', help='complex128 / float64')
ap.add_argument('--dist', default='uniform', help='uniform | radial | radial-ordered (2-D)')
ap.add_argument('--one-call', action='store_true', help='nufft_hip_execute_with_points instead of set_points + execute')
a = ap.parse_args()
grid = [int(g) for g in a.grid.split(',')]; M = int(a.M); rank = len(grid)
g = torch.Generator(device='cuda').manual_seed(2)
pts = (torch.rand((M, rank), generator=g, device='cuda') * 2 - 1) * np.pi
if a.dist == 'radial':
  r = torch.rand(M, generator=g, device='cuda') * np.pi; th = torch.rand(M, generator=g, device='cuda') * 2 * np.pi
  pts = torch.stack([r * torch.cos(th), r * torch.sin(th)], dim=1)
elif a.dist == 'radial-ordered':   # spokes through the centre, stored spoke after spoke
  ns = 1000; ang = torch.arange(M // ns, device='cuda') * (np.pi * 0.6180339887); s = torch.linspace(-np.pi, np.pi, ns + 1, device='cuda')[:ns]
  pts = torch.stack([(s[None, :] * torch.cos(ang)[:, None]).reshape(-1), (s[None, :] * torch.sin(ang)[:, None]).reshape(-1)], dim=1)
  M = pts.shape[0]
lead = [a.ntransf] if a.ntransf > 1 else []
if a.type == 'type_1':
  src = torch.complex(torch.rand(lead + [M], generator=g, device='cuda') - .5, torch.rand(lead + [M], generator=g, device='cuda') - .5)
else:
  src = torch.complex(torch.rand(lead + grid, generator=g, device='cuda') - .5, torch.rand(lead + grid, generator=g, device='cuda') - .5)
if a.double: pts = pts.double(); src = src.to(torch.complex128)
from tensorflow_nufft import _lib
tuning = 0
for b in filter(None, a.tuning.split(',')): tuning |= _lib.TUNE[b]
plan = tfft.Plan(a.type, grid, 'forward', num_transforms=a.ntransf, tol=a.tol, dtype=torch.complex128 if a.double else torch.complex64, spread_method=a.method, max_subproblem_size=a.S, lds_accumulate=a.acc, tuning=tuning)
for _ in range(a.steps):
  if a.one_call:
    out = plan.execute_with_points(pts, src)
  else:
    plan.set_points(pts)
    out = plan.execute(src)
torch.cuda.synchronize()
print('done', out.shape)
