#!/usr/bin/env python3
"""A/B of the stack form of the 3-D w = 7 / 8 fixed-point spreader (options.tuning STACK_ON / STACK_OFF), one run:
HIP-event stage times per call (set_points + execute), the difference of the two outputs, stack statistics.

    python tools/ab_stack.py [--cases 128:8e5,256:3e7,256:1e8] [--tol 1e-6] [--sweep]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.environ.get('NUFFT_PKG', os.path.join(ROOT, 'tensorflow-nufft_amd')))   # (NUFFT_PKG: a variant build, tools/variant_build.sh)
import numpy as np
import torch
import tensorflow_nufft as tfft
from tensorflow_nufft._lib import TUNE


def run(plan, pts, c, steps):
  for _ in range(2):
    plan.set_points(pts); out = plan.execute(c)
  plan.set_timing(True); plan.get_timing()
  for _ in range(steps):
    plan.set_points(pts); out = plan.execute(c)
  tm = plan.get_timing()
  st = {k: v[0] / steps * 1e3 for k, v in tm.items() if v[1]}   # (per call: a stage may run more than once per call)
  return out, st


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--cases', default='128:8e5,256:3e7,256:1e8')
  ap.add_argument('--tol', type=float, default=1e-6)
  ap.add_argument('--steps', type=int, default=5)
  ap.add_argument('--sweep', action='store_true')
  ap.add_argument('--dist', default='uniform')
  ap.add_argument('--only', default='', help='STACK_ON | STACK_OFF: just that variant')
  args = ap.parse_args()
  for case in args.cases.split(','):
    n, M = case.split(':'); n = int(n); M = int(float(M))
    grid = [n, n, n]
    g = torch.Generator(device='cuda').manual_seed(1)
    if args.dist == 'uniform':
      pts = (torch.rand((M, 3), generator=g, device='cuda') * 2 - 1) * np.pi
    else:   # gaussian cluster
      pts = (torch.randn((M, 3), generator=g, device='cuda') * 0.4).clamp(-np.pi, np.pi)
    c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
    print(f'# {n}^3 modes, M = {M:.3g} ({M / (2 * n) ** 3:.3f} per fine cell), tol {args.tol:g}, {args.dist} points', flush=True)
    ref = None
    variants = [('STACK_OFF', TUNE['STACK_OFF'], 0, 0), ('STACK_ON', TUNE['STACK_ON'], 0, 0)]
    if args.only:
      variants = [v for v in variants if v[0] == args.only]
    if args.sweep:
      variants += [(f'STACK_ON len={l} cap={cp}', TUNE['STACK_ON'], l, cp) for l in (2, 4, 8, 16, 32) for cp in (4096, 8192, 16384)]
    for name, tune, ln, cp in variants:
      plan = tfft.Plan('type_1', grid, 'forward', tol=args.tol, tuning=tune)
      if ln or cp:
        plan.stack_params(ln, cp)
      out, st = run(plan, pts, c, args.steps)
      total = sum(st.values())
      b = plan.sub_bounds()
      live = b[b != 0]
      line = f'{name:28s} ' + ' '.join(f'{k}={v:.0f}us' for k, v in st.items()) + f' | all {total / 1e3:.3f} ms'
      if live.size:
        line += f' | {live.size} bounds, B mean {np.abs(live).mean():.1f} max {np.abs(live).max():.1f}, {int((live < 0).sum())} flagged'
      sk = plan.stacks()
      if sk.size:
        nz = sk[:, 1] >> 16
        line += f' | stacks {sk.shape[0]}, tiles per stack mean {nz.mean():.1f} max {nz.max()}, pieces {int((sk[:, 2] >= 0).sum())}'
      if ref is None:
        ref = out
      else:
        line += f' | rel-l2 vs first {float(torch.linalg.norm(out - ref) / torch.linalg.norm(ref)):.2e}'
      print(line, flush=True)
      plan.close()
      torch.cuda.empty_cache()


if __name__ == '__main__':
  main()
