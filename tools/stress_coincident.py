# Adversarial point sets for the accumulation paths: every point at (nearly) one spot, type 1, float and double,
# against the fp64 oracle at tol 1e-12 and at the same tol.
import os, sys
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
from oracle import oracle
rng = np.random.default_rng(7)
for grid, tols in (([300], (1e-6, 1e-4)), ([96, 80], (1e-6, 1e-4, 1e-2)), ([1024, 1024], (1e-6,)), ([32, 32, 32], (1e-6, 1e-5, 1e-3))):
  rank = len(grid)
  for M in (60000, 600000):
    for kind in ('coincident', 'one_cell'):
      if kind == 'coincident':
        pts = np.where(rng.integers(0, 2, (M, rank)) == 1, np.pi, -np.pi) + rng.uniform(-1e-3, 1e-3, (M, rank))
      else:   # spread over one fine cell: same stencil start, different kernel arguments
        pts = 0.3 + rng.uniform(0, 1, (M, rank)) * (2 * np.pi / (2 * np.array(grid)))
      pts32 = pts.astype(np.float32)
      c = (rng.uniform(-.5, .5, M) + 1j * rng.uniform(-.5, .5, M)).astype(np.complex64)
      truth = oracle.nufft(c.astype(np.complex128), pts32, grid, 'type_1', 'forward', tol=1e-12, sigma=2.0)
      row = []
      for tol in tols:
        same = oracle.nufft(c.astype(np.complex128), pts32, grid, 'type_1', 'forward', tol=tol, sigma=2.0)
        out = tfft.nufft(torch.from_numpy(c).cuda(), torch.from_numpy(pts32).cuda(), grid_shape=grid, transform_type='type_1', tol=tol).cpu().numpy()
        e, r = np.linalg.norm(out - truth) / np.linalg.norm(truth), np.linalg.norm(same - truth) / np.linalg.norm(truth)
        row.append(f"tol {tol:g}: gpu {e:.2e} oracle {r:.2e}{' <<<' if e > max(tol, 1.3 * r) else ''}")
      print(grid, M, kind, ' | '.join(row))
