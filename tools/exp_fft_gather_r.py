"""r06 experiment: lines per workgroup of the type-2 (gather) power-of-two FFT passes on large 3-D grids (experiment
build with -DNUFFT_MIX_SHAPE_ENV, NUFFT_FFT_R_GATHER / NUFFT_FFT_R from the environment)."""
import os, sys
pkg = os.environ.get('NUFFT_PKG')
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, pkg if pkg else os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
from tensorflow_nufft import _lib
print(_lib.lib().nufft_hip_build_info().decode())
def fft_us(ttype, grid, M=1_000_000, steps=5):
  g = torch.Generator(device='cuda').manual_seed(1)
  pts = (torch.rand((M, len(grid)), generator=g, device='cuda') * 2 - 1) * np.pi
  shp = [M] if ttype == 'type_1' else grid
  src = torch.complex(torch.rand(shp, generator=g, device='cuda'), torch.rand(shp, generator=g, device='cuda'))
  plan = tfft.Plan(ttype, grid, 'forward', tol=1e-6)
  plan.set_points(pts)
  for _ in range(2): plan.execute(src)
  plan.set_timing(True); plan.get_timing()
  for _ in range(steps): plan.execute(src)
  tm = plan.get_timing()
  plan.close()
  return tm['fft'][0] / max(tm['fft'][1], 1) * 1e3
for grid in ([512] * 3, [256] * 3, [1024, 1024], [128] * 3):
  for tt, var in (('type_2', 'NUFFT_FFT_R_GATHER'), ('type_1', 'NUFFT_FFT_R')):
    row = []
    for r in (0, 32, 16, 8, 4, 2):
      if r: os.environ[var] = str(r)
      else: os.environ.pop(var, None)
      try: row.append(f'{fft_us(tt, grid):9.1f}')
      except Exception as e: row.append('      n/a')
    os.environ.pop(var, None)
    print(f'{"x".join(map(str, grid)):15} {tt[-1]}  R=auto,32,16,8,4,2: ' + ' '.join(row), flush=True)
