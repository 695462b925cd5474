"""r06: whole transforms and the FFT stage on fine grids that are not powers of two (the sizes the reference's
next_smooth_int picks, nufft_util.cc:119-133) beside their power-of-two neighbours. `--rocfft`: also force
rocFFT + deconvolve (tuning ROCFFT) on every case, for the same-run A/B."""
import os, sys, time, argparse
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
from tensorflow_nufft._lib import TUNE

def rnd_c(shape, g, dtype):
  rt = torch.float32 if dtype == torch.complex64 else torch.float64
  return torch.complex(torch.rand(shape, generator=g, device='cuda', dtype=rt) - .5,
                       torch.rand(shape, generator=g, device='cuda', dtype=rt) - .5)

def run(ttype, grid, M, tol, steps, dtype=torch.complex64, **kw):
  g = torch.Generator(device='cuda').manual_seed(3)
  rank = len(grid)
  rt = torch.float32 if dtype == torch.complex64 else torch.float64
  pts = (torch.rand((M, rank), generator=g, device='cuda', dtype=rt) * 2 - 1) * np.pi
  plan = tfft.Plan(ttype, grid, 'forward', tol=tol, dtype=dtype, **kw)
  src = rnd_c([M] if ttype == 'type_1' else grid, g, dtype)
  def step():
    plan.set_points(pts); plan.execute(src)
  step(); step()
  plan.set_timing(True); plan.get_timing()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(steps): step()
  torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
  tm = plan.get_timing()
  i = plan.info()
  st = {k: v[0] / max(v[1], 1) * 1e3 for k, v in tm.items() if v[1]}
  fft_us = st.get('fft', 0.0) + st.get('deconvolve', 0.0)
  nf = list(i.fine_dims) if hasattr(i, 'fine_dims') else []
  cells = float(np.prod(nf)) if nf else float('nan')
  tag = 'x'.join(str(n) for n in grid)
  print(f'{ttype[-1]} {tag:>14} nf={nf} M={M:.0e} tol={tol:g} {"c128" if dtype == torch.complex128 else "c64"} '
        f'{"tuning=%x" % kw.get("tuning") if kw.get("tuning") else "default "}: {dt*1e3:8.3f} ms/step {dt/M*1e9:7.3f} ns/pt | fft+deconv {fft_us:8.1f} us '
        f'= {fft_us*1e3/cells:6.3f} ns/cell | ' + ' '.join(f'{k}={v:.0f}' for k, v in st.items()), flush=True)
  plan.close(); del pts, src; torch.cuda.empty_cache()

if __name__ == '__main__':
  ap = argparse.ArgumentParser()
  ap.add_argument('--rocfft', action='store_true')
  ap.add_argument('--quick', action='store_true')
  ap.add_argument('--types', default='12')
  a = ap.parse_args()
  cases3 = [[256] * 3, [240] * 3, [200] * 3, [192] * 3, [128] * 3, [120] * 3, [96] * 3, [320] * 3]
  cases2 = [[1024] * 2, [960] * 2, [1000] * 2, [768] * 2, [1280] * 2, [512] * 2, [480] * 2]
  if a.quick: cases3, cases2 = cases3[:4], cases2[:3]
  print(torch.cuda.get_device_name(0))
  for t in a.types:
    tt = 'type_' + t
    for grid in cases3:
      for M in ([10_000_000] if a.quick else [10_000_000, 30_000_000]):
        run(tt, grid, M, 1e-6, 5)
        if a.rocfft: run(tt, grid, M, 1e-6, 5, tuning=TUNE['ROCFFT'])
    for grid in cases2:
      run(tt, grid, 10_000_000, 1e-6, 10)
      if a.rocfft: run(tt, grid, 10_000_000, 1e-6, 10, tuning=TUNE['ROCFFT'])
  # double precision, two sizes
  for grid in ([240] * 3, [256] * 3):
    run('type_1', grid, 10_000_000, 1e-9, 3, dtype=torch.complex128)
    if a.rocfft: run('type_1', grid, 10_000_000, 1e-9, 3, dtype=torch.complex128, tuning=TUNE['ROCFFT'])
