#!/usr/bin/env python3
"""The eight cases of the reference's own benchmark harness (NUFFTOpsBenchmark.benchmark_nufft,
tensorflow_nufft/python/ops/nufft_ops_test.py:728-809; it prints at run time and publishes
nothing) through `tfft.nufft` on the GPU: 2 burn-in calls, 50 timed calls, wall time per call;
beside it the CPU oracle (port of the reference CPU path) on the same inputs, 3 calls."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
from oracle import oracle

CASES = [   # source_shape, points_shape, transform_type, grid_shape   (nufft_ops_test.py:732-741)
    ([256, 256], [200000, 2], 'type_2', None),
    ([16, 256, 256], [200000, 2], 'type_2', None),
    ([16, 256, 256], [16, 200000, 2], 'type_2', None),
    ([200000], [200000, 2], 'type_1', [256, 256]),
    ([16, 200000], [200000, 2], 'type_1', [256, 256]),
    ([16, 200000], [16, 200000, 2], 'type_1', [256, 256]),
    ([128, 128, 128], [800000, 3], 'type_2', None),
    ([800000], [800000, 3], 'type_1', [128, 128, 128]),
]
rng = np.random.default_rng(0)
rnd = lambda shape: rng.random(shape, dtype=np.float32) - 0.5
rows = []
for source_shape, points_shape, ttype, grid in CASES:       # GPU timings first (the oracle's OpenMP team would disturb them)
  src = (rnd(source_shape) + 1j * rnd(source_shape)).astype(np.complex64)
  pts = (rnd(points_shape) * 2.0 * np.pi).astype(np.float32)
  s, p = torch.from_numpy(src).cuda(), torch.from_numpy(pts).cuda()
  for _ in range(2):
    out = tfft.nufft(s, p, grid_shape=grid, transform_type=ttype)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(50):
    out = tfft.nufft(s, p, grid_shape=grid, transform_type=ttype)
  torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 50 * 1e3
  elem_rank = 1 if ttype == 'type_1' else len(points_shape[-1:]) and points_shape[-1]
  ntransf = int(np.prod(source_shape[:len(source_shape) - elem_rank])) if len(source_shape) > elem_rank else 1
  npts = points_shape[-2] * max(ntransf, int(np.prod(points_shape[:-2])) if len(points_shape) > 2 else 1)
  rows.append((source_shape, points_shape, ttype, grid, ms, npts, src, pts, out.cpu().numpy()))
print('# source_shape | points_shape | type | grid | GPU ms per call (tfft.nufft, complex64, tol 1e-6) | M points x transforms per s (Mpts/s) | CPU oracle ms (threads) | rel-l2 GPU vs oracle(fp64, tol 1e-12)')
for source_shape, points_shape, ttype, grid, ms, npts, src, pts, outn in rows:
  p0 = pts if pts.ndim == 2 else pts[0]
  s0 = src if pts.ndim == 2 else src[0]
  nth = oracle.default_threads()   # (the container's cgroup CPU quota, not the host's hardware threads: EXPERIMENTS.md 10.12)
  t1 = time.perf_counter()
  for _ in range(3):
    oracle.nufft(s0, p0, grid, ttype, 'forward', tol=1e-6, kerevalmeth=1, nthreads=nth)
  cpu_ms = (time.perf_counter() - t1) / 3 * 1e3 * (1 if pts.ndim == 2 else pts.shape[0])
  truth = oracle.nufft(s0.astype(np.complex128), p0, grid, ttype, 'forward', tol=1e-12, sigma=2.0)
  got = outn if pts.ndim == 2 else outn[0]
  err = np.linalg.norm(got - truth) / np.linalg.norm(truth)
  print(f'{source_shape} | {points_shape} | {ttype} | {grid} | {ms:.3f} | {npts / ms / 1e3:.0f} | {cpu_ms:.1f} ({nth}) | {err:.1e}')
