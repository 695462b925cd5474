"""Whole-output error at tol 1e-6 when 1e6-2e6 points sit within a few cells of one spot (hundreds of subproblems of
one tile adding float partial sums to the same fine-grid cells), 2-D and 1-D. EXPERIMENTS.md section 10.14.
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
from oracle import oracle
rng = np.random.default_rng(1)
for grid, M in (([64, 64], 1_000_000), ([1024, 1024], 2_000_000), ([300], 1_000_000)):
  r = len(grid)
  pts = (rng.uniform(-2.5, 2.5, (1, r)) + 0.003 * rng.standard_normal((M, r))).astype(np.float32)
  c = (rng.standard_normal(M) + 1j * rng.standard_normal(M)).astype(np.complex64)
  truth = oracle.nufft(c.astype(np.complex128), pts, grid, 'type_1', 'forward', tol=1e-12, sigma=2.0)
  same = oracle.nufft(c.astype(np.complex128), pts, grid, 'type_1', 'forward', tol=1e-6, sigma=2.0)
  out = tfft.nufft(torch.from_numpy(c).cuda(), torch.from_numpy(pts).cuda(), grid_shape=grid, transform_type='type_1', tol=1e-6).cpu().numpy()
  e = lambda a: np.linalg.norm(a - truth) / np.linalg.norm(truth)
  print(grid, M, 'err %.3e  reference rule %.3e' % (e(out), e(same)), flush=True)
