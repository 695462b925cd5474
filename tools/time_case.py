import os, sys, time
ROOT = os.environ['GRAFT_REPO_ROOT']
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
rng = np.random.default_rng(0)
M = int(float(sys.argv[1])); grid = [int(g) for g in sys.argv[2].split(',')]
pts = torch.from_numpy(((rng.random((M, len(grid)), dtype=np.float32) - .5) * 2 * np.pi)).cuda()
c = torch.from_numpy((rng.random(M, dtype=np.float32) - .5 + 1j * (rng.random(M, dtype=np.float32) - .5)).astype(np.complex64)).cuda()
for _ in range(3): out = tfft.nufft(c, pts, grid_shape=grid, transform_type='type_1')
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): out = tfft.nufft(c, pts, grid_shape=grid, transform_type='type_1')
torch.cuda.synchronize(); print('tfft.nufft ms/call', (time.perf_counter() - t0) / 50 * 1e3)
plan = tfft.Plan('type_1', grid)
for _ in range(3): plan.execute_with_points(pts, c)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): plan.execute_with_points(pts, c)
torch.cuda.synchronize(); print('plan one-call ms/call', (time.perf_counter() - t0) / 50 * 1e3)
t0 = time.perf_counter()
for _ in range(50): plan.set_points(pts); plan.execute(c)
torch.cuda.synchronize(); print('plan two-call ms/call', (time.perf_counter() - t0) / 50 * 1e3)
