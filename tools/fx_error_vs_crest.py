#!/usr/bin/env python3
"""How much error does the packed fixed-point accumulation of the 3-D w = 8 spreader add, as a function of the
count-filter bound B (point density) and of the strengths' crest factor (largest / rms)?

3-D type 1 on 64^3 modes (128^3 fine cells), tol 1e-6, uniform points at several densities, five strength laws;
error against the fp64 oracle (sigma 2, tol 1e-12) with the fixed-point kernel (lds_accumulate 0) and with the
fp64 planes (lds_accumulate 1); "added" = their difference in quadrature. Run on a library built with
-DNUFFT_FX_BOUND_LIMIT=1e9 (tools/fx_error_vs_crest.sh) to see bounds above the product's limit too.

    python tools/fx_error_vs_crest.py [--lib path/to/libnufft_hip.so]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--lib', default=None)
  ap.add_argument('--grid', type=int, default=64)
  ap.add_argument('--densities', default='0.25,0.75,1.5,3')
  ap.add_argument('--tol', type=float, default=1e-6)
  args = ap.parse_args()
  import tensorflow_nufft._lib as L
  if args.lib:
    L.LIB_PATH = os.path.abspath(args.lib)
  import torch
  import tensorflow_nufft as tfft
  from oracle import oracle
  n = args.grid
  grid = [n, n, n]
  rng = np.random.default_rng(7)
  print(f'# 3-D type 1, {n}^3 modes, tol {args.tol:g}; library {L.LIB_PATH}')
  print('# density  strengths        crest(top/rms)  B mean / max   fp64-plane subproblems | err fp64 planes  err fixed point  added (quadrature)  added / (B mean x crest)')
  for d in [float(x) for x in args.densities.split(',')]:
    M = int(d * (2 * n) ** 3)
    pts = rng.uniform(-np.pi, np.pi, (M, 3)).astype(np.float32)
    z = rng.standard_normal(M) + 1j * rng.standard_normal(M)
    laws = {
        'uniform': rng.uniform(-.5, .5, M) + 1j * rng.uniform(-.5, .5, M),
        'gaussian': z,
        'lognormal s=1': z * np.exp(rng.standard_normal(M)),
        'lognormal s=2': z * np.exp(2 * rng.standard_normal(M)),
        'six decades': z * 10.0 ** rng.uniform(-3, 3, M),
    }
    dp = torch.from_numpy(pts).cuda()
    for name, c in laws.items():
      c = c.astype(np.complex64)
      m = np.maximum(np.abs(c.real), np.abs(c.imag))
      crest = float(m.max() / np.sqrt((m.astype(np.float64) ** 2).mean()))
      truth = oracle.nufft(c.astype(np.complex128), pts, grid, 'type_1', 'forward', tol=1e-12, sigma=2.0)
      den = np.linalg.norm(truth)
      errs = {}
      bstat = ''
      for mode in (1, 0):
        plan = tfft.Plan('type_1', grid, 'forward', tol=args.tol, lds_accumulate=mode)
        plan.set_points(dp)
        out = plan.execute(torch.from_numpy(c).cuda()).cpu().numpy()
        if mode == 0:
          b = plan.sub_bounds()
          live = b[b != 0]
          bm = float(np.abs(live).mean()) if live.size else 0.0
          bstat = f'{bm:6.1f} / {np.abs(live).max() if live.size else 0:6.1f}   {int((live < 0).sum()):5d} of {live.size:5d}'
        plan.close()
        errs[mode] = np.linalg.norm(out - truth) / den
      added = np.sqrt(max(errs[0] ** 2 - errs[1] ** 2, 0.0))
      print(f'{d:7.2f}  {name:15s} {crest:10.2f}       {bstat} | {errs[1]:.3e}  {errs[0]:.3e}  {added:.3e}  {added / max(bm * crest, 1e-30):.2e}', flush=True)


if __name__ == '__main__':
  main()
