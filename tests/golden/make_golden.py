"""Generates the committed fixtures under tests/golden/. Run in the build
container (needs /root/reference for the ref_*.npz files):

    python tests/golden/make_golden.py

Fixtures are DATA only (inputs + expected outputs):
  ref_horner.npz    outputs of the reference's generated Horner kernel tables
                    (kernel_horner_sigma2.inc, _sigma125.inc, _sigma2_gpu.inc,
                    compiled from /root/reference by oracle/Makefile) at fixed
                    offsets -- pins the ES kernel arithmetic.
  ref_legendre.npz  outputs of the reference's Gauss-Legendre rule
                    (legendre_rule_fast.cc) -- pins the quadrature.
  nudft_cases.npz   seeded random inputs + float64 dense-NUDFT outputs on the
                    shapes the reference's own tests use
                    (nufft_ops_test.py:87-221, 351-417): [8], [6,8], [4,8,6],
                    24x24, both types, both signs. The reference draws inputs
                    with TF's RNG (not reproducible without TF), so the same
                    distributions are drawn with numpy default_rng(seed).
  nudft_mid.npz     a mid-size 2D/3D/1D set (64x64, M=6000, etc.).
"""
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, '..', '..')))
from oracle import oracle  # noqa: E402


def _p(a):
  return a.ctypes.data_as(ctypes.c_void_p)


def make_ref():
  oracle.build()
  ref = oracle.ref_lib()
  if ref is None:
    print('reference pieces not built; skipping ref_*.npz')
    return
  out = {}
  for w in range(2, 17):
    x1 = np.linspace(-w / 2, -w / 2 + 1, 33)
    out[f'x1_w{w}'] = x1
    for sigma, tag in ((2.0, 's2'), (1.25, 's125')):
      ker = np.zeros((33, 20))
      ref.ref_horner_f64(33, _p(x1), w, ctypes.c_double(sigma), _p(ker), 20)
      out[f'ker_{tag}_w{w}'] = ker[:, :w].copy()
    ker = np.zeros((33, 20))
    ref.ref_horner_gpu_f64(33, _p(x1), w, _p(ker), 20)
    out[f'ker_gpu_w{w}'] = ker[:, :w].copy()
  np.savez_compressed(os.path.join(HERE, 'ref_horner.npz'), **out)
  out = {}
  for w in range(2, 17):
    q = int(2 + 3.0 * w / 2)
    z = np.zeros(2 * q)
    wt = np.zeros(2 * q)
    ref.ref_legendre_glr(2 * q, _p(z), _p(wt))
    out[f'z_{2*q}'] = z
    out[f'w_{2*q}'] = wt
  np.savez_compressed(os.path.join(HERE, 'ref_legendre.npz'), **out)


def _case(rng, grid, M, tt, fd):
  rank = len(grid)
  pts = rng.uniform(-np.pi, np.pi, (M, rank))
  if tt == 'type_1':
    src = rng.uniform(-.5, .5, M) + 1j * rng.uniform(-.5, .5, M)
  else:
    src = rng.uniform(-.5, .5, grid) + 1j * rng.uniform(-.5, .5, grid)
  # inputs are stored at float32 resolution so f32 and f64 runs see the same data
  pts = pts.astype(np.float32).astype(np.float64)
  src = src.astype(np.complex64).astype(np.complex128)
  out = oracle.nudft(src, pts, grid, tt, fd)
  return pts, src, out


def make_nudft():
  rng = np.random.default_rng(0)
  out = {}
  names = []
  for grid in ([8], [6, 8], [4, 8, 6], [24, 24]):
    M = int(np.prod(grid))
    for tt in ('type_1', 'type_2'):
      for fd in ('forward', 'backward'):
        name = f"g{'x'.join(map(str, grid))}_{tt}_{fd}"
        pts, src, res = _case(rng, grid, M, tt, fd)
        out[name + '_points'] = pts.astype(np.float32)
        out[name + '_source'] = src.astype(np.complex64)
        out[name + '_target'] = res
        names.append(name)
  out['names'] = np.array(names)
  np.savez_compressed(os.path.join(HERE, 'nudft_cases.npz'), **out)

  rng = np.random.default_rng(1)
  out = {}
  names = []
  for grid, M in (([200], 3000), ([64, 64], 6000), ([16, 20, 24], 5000),
                  ([15, 30], 2000)):
    for tt in ('type_1', 'type_2'):
      fd = 'forward' if tt == 'type_1' else 'backward'
      name = f"g{'x'.join(map(str, grid))}_{tt}_{fd}"
      pts, src, res = _case(rng, grid, M, tt, fd)
      out[name + '_points'] = pts.astype(np.float32)
      out[name + '_source'] = src.astype(np.complex64)
      out[name + '_target'] = res
      names.append(name)
  out['names'] = np.array(names)
  np.savez_compressed(os.path.join(HERE, 'nudft_mid.npz'), **out)


if __name__ == '__main__':
  make_ref()
  make_nudft()
  for f in sorted(os.listdir(HERE)):
    if f.endswith('.npz'):
      print(f, os.path.getsize(os.path.join(HERE, f)))
