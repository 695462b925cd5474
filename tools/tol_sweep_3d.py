#!/usr/bin/env python3
"""3-D type-1 tolerance sweep on 256^3 modes (512^3 fine cells): HIP-event stage times per call
(set_points + execute), and for tol <= 1e-5 the whole-output error against the fp64 oracle (sigma = 2,
tol 1e-12) with the r04 fixed-point kernel (spread_patch3_kernel) and with the r03 kernels
(options.tuning FXPATCH_OFF) in the same run.

    python tools/tol_sweep_3d.py [--M 30000000] [--tols 1e-4,1e-5,1e-6] [--no-oracle]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np
import torch
import tensorflow_nufft as tfft
from tensorflow_nufft._lib import TUNE


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--M', type=int, default=30_000_000)
  ap.add_argument('--grid', type=int, default=256)
  ap.add_argument('--tols', default='1e-1,1e-2,1e-3,1e-4,1e-5,1e-6')
  ap.add_argument('--no-oracle', action='store_true')
  ap.add_argument('--steps', type=int, default=3)
  ap.add_argument('--strengths', default='uniform', help='uniform | gaussian | lognormal (exp of a unit normal times a complex normal)')
  args = ap.parse_args()
  n, M = args.grid, args.M
  grid = [n, n, n]
  print(f'# 3-D type 1, {n}^3 modes ({2*n}^3 fine cells), M = {M:.3g} uniform points ({M / (2*n)**3:.2f} per fine cell), '
        f'complex64: HIP-event stage times per call (set_points + execute)')
  g = torch.Generator(device='cuda').manual_seed(1)
  pts = (torch.rand((M, 3), generator=g, device='cuda') * 2 - 1) * np.pi
  if args.strengths == 'uniform':
    c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
  else:
    c = torch.complex(torch.randn(M, generator=g, device='cuda'), torch.randn(M, generator=g, device='cuda'))
    if args.strengths == 'lognormal':
      c = c * torch.exp(torch.randn(M, generator=g, device='cuda'))
  print(f'# strengths: {args.strengths}')
  truth = None
  for tol in [float(t) for t in args.tols.split(',')]:
    variants = [('r05', 0), ('per subproblem (STACK_OFF)', TUNE['STACK_OFF'])]
    if tol <= 2e-5:
      variants.append(('r03 kernels (FXPATCH_OFF)', TUNE['FXPATCH_OFF']))
    for name, tune in variants:
      plan = tfft.Plan('type_1', grid, 'forward', tol=tol, tuning=tune)
      for _ in range(2):
        plan.set_points(pts); out = plan.execute(c)
      plan.set_timing(True); plan.get_timing()
      for _ in range(args.steps):
        plan.set_points(pts); out = plan.execute(c)
      tm = plan.get_timing()
      i = plan.info()
      line = (f'tol {tol:g} w={i.kernel_width} tile={list(i.tile_dims)} sub<={i.max_subproblem_size} [{name}]: ' +
              ' '.join(f'{k}={v[0] / args.steps * 1e3:.0f}us' for k, v in tm.items() if v[1]))
      total = sum(v[0] / args.steps for v in tm.values() if v[1])
      line += f' | all stages {total:.2f} ms'
      b = plan.sub_bounds()
      if b.size:
        live = b[b != 0]
        line += (f' | bounds: {live.size} {"stacks" if plan.stacks().size else "subproblems"}, B mean {np.abs(live).mean():.1f} max {np.abs(live).max():.1f}, '
                 f'{int((live < 0).sum())} on fp64 planes')
      if tol <= 2e-5 and not args.no_oracle:
        if truth is None:
          from oracle import oracle
          truth = oracle.nufft(c.cpu().numpy().astype(np.complex128), pts.cpu().numpy(), grid, 'type_1', 'forward',
                               tol=1e-12, sigma=2.0)
        o = out.cpu().numpy()
        err = np.linalg.norm(o - truth) / np.linalg.norm(truth)
        line += f' | rel-l2 vs fp64 oracle {err:.3e}'
      print(line, flush=True)
      plan.close()
      torch.cuda.empty_cache()


if __name__ == '__main__':
  main()
