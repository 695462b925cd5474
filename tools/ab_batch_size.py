import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
rng = np.random.default_rng(0)
for (grid, M, B) in (([256,256], 200000, 16), ([512,512], 1000000, 32)):
  pts = torch.from_numpy(((rng.random((M, 2), dtype=np.float32) - .5) * 2 * np.pi)).cuda()
  c = torch.from_numpy((rng.random((B, M), dtype=np.float32) - .5 + 1j * (rng.random((B, M), dtype=np.float32) - .5)).astype(np.complex64)).cuda()
  f = torch.from_numpy((rng.random([B] + grid, dtype=np.float32) - .5 + 1j * (rng.random([B] + grid, dtype=np.float32) - .5)).astype(np.complex64)).cuda()
  for mb in (0, 4, 8, 16, 32):
    opt = tfft.Options(max_batch_size=mb) if mb else None
    for tt, src, kw in (('type_1', c, dict(grid_shape=grid)), ('type_2', f, {})):
      for _ in range(3): out = tfft.nufft(src, pts, transform_type=tt, options=opt, **kw)
      torch.cuda.synchronize(); t0 = time.perf_counter()
      for _ in range(30): out = tfft.nufft(src, pts, transform_type=tt, options=opt, **kw)
      torch.cuda.synchronize(); print(grid, M, B, 'max_batch', mb, tt, f'{(time.perf_counter()-t0)/30*1e3:.3f} ms')
