#!/usr/bin/env python3
"""A/B of the 3-D interpolation over stacks of tiles (options.tuning STACK_ON / STACK_OFF), type 2, one run:
HIP-event stage times per call (set_points + execute) and the difference of the two outputs.

    python tools/ab_stack_interp.py [--cases 128:8e5,256:3e6,...] [--tol 1e-6] [--double]
"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.environ.get('NUFFT_PKG', os.path.join(ROOT, 'tensorflow-nufft_amd')))
import numpy as np, torch
import tensorflow_nufft as tfft
from tensorflow_nufft._lib import TUNE
ap = argparse.ArgumentParser()
ap.add_argument('--cases', default='128:8e5,256:3e6,256:1e7,256:3e7')
ap.add_argument('--tol', type=float, default=1e-6)
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--double', action='store_true')
args = ap.parse_args()
cdt = torch.complex128 if args.double else torch.complex64
for case in args.cases.split(','):
  n, M = case.split(':'); n = int(n); M = int(float(M))
  grid = [n, n, n]
  g = torch.Generator(device='cuda').manual_seed(1)
  pts = (torch.rand((M, 3), generator=g, device='cuda') * 2 - 1) * np.pi
  f = torch.complex(torch.rand(grid, generator=g, device='cuda') - .5, torch.rand(grid, generator=g, device='cuda') - .5).to(cdt)
  if args.double: pts = pts.double()
  print(f'# type 2, {n}^3 modes, M = {M:.3g} ({M / (2 * n) ** 3:.3f} per fine cell), tol {args.tol:g}, {"complex128" if args.double else "complex64"}', flush=True)
  ref = None
  for name in ('STACK_OFF', 'STACK_ON'):
    plan = tfft.Plan('type_2', grid, 'forward', tol=args.tol, tuning=TUNE[name], dtype=cdt)
    for _ in range(2):
      plan.set_points(pts); out = plan.execute(f)
    plan.set_timing(True); plan.get_timing()
    for _ in range(args.steps):
      plan.set_points(pts); out = plan.execute(f)
    tm = plan.get_timing()
    st = {k: v[0] / args.steps * 1e3 for k, v in tm.items() if v[1]}   # (per call: a stage may run more than once per call)
    i = plan.info()
    line = f'{name:10s} w={i.kernel_width} tile={list(i.tile_dims)} ' + ' '.join(f'{k}={v:.0f}us' for k, v in st.items()) + f' | all {sum(st.values()) / 1e3:.3f} ms'
    sk = plan.stacks()
    if sk.size:
      line += f' | stacks {sk.shape[0]}, tiles per stack {(sk[:, 1] >> 16).mean():.1f}'
    if ref is None: ref = out
    else: line += f' | rel-l2 vs first {float(torch.linalg.norm(out - ref) / torch.linalg.norm(ref)):.2e}'
    print(line, flush=True)
    plan.close()
    torch.cuda.empty_cache()
