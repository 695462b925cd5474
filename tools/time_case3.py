# A/B timing of one transform through tfft.nufft: python tools/time_case3.py M n0[,n1[,n2]] type tol [c64|c128]
import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
rng = np.random.default_rng(0)
M = int(float(sys.argv[1])); grid = [int(g) for g in sys.argv[2].split(',')]
ttype = sys.argv[3] if len(sys.argv) > 3 else 'type_1'
tol = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-6
dbl = len(sys.argv) > 5 and sys.argv[5] == 'c128'
pts = torch.from_numpy(((rng.random((M, len(grid)), dtype=np.float32) - .5) * 2 * np.pi)).cuda()
if dbl: pts = pts.double()
if ttype == 'type_1':
  src = torch.from_numpy((rng.random(M, dtype=np.float32) - .5 + 1j * (rng.random(M, dtype=np.float32) - .5)).astype(np.complex64)).cuda()
  kw = dict(grid_shape=grid)
else:
  src = torch.from_numpy((rng.random(grid, dtype=np.float32) - .5 + 1j * (rng.random(grid, dtype=np.float32) - .5)).astype(np.complex64)).cuda()
  kw = {}
if dbl: src = src.to(torch.complex128)
n = 20 if M >= 5e7 else 50
for _ in range(3): out = tfft.nufft(src, pts, transform_type=ttype, tol=tol, **kw)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(n): out = tfft.nufft(src, pts, transform_type=ttype, tol=tol, **kw)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(f'{ttype} {"c128" if dbl else "c64"} M={M:.0e} grid={grid} tol={tol:g}: {ms:.3f} ms/call = {ms * 1e6 / M:.2f} ns/pt  |out|={float(out.abs().sum()):.6e}')
