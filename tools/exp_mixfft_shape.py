"""r06 experiment: lines per workgroup (R) of the mixed-radix FFT passes, forced through NUFFT_MIX_R in an experiment
build (tools/variant_build.sh mixshape nufft_fft.hip -DNUFFT_MIX_SHAPE_ENV; NUFFT_PKG=/tmp/variants/mixshape).
FFT stage (HIP events of the plan) of type-1 transforms over fine-grid sizes; the point count is small so that the
FFT stage is not disturbed by the spreader's tail."""
import os, sys
pkg = os.environ.get('NUFFT_PKG')
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, pkg if pkg else os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
from tensorflow_nufft import _lib
print(_lib.lib().nufft_hip_build_info().decode())

def fft_us(ttype, grid, M=1_000_000, steps=20):
  g = torch.Generator(device='cuda').manual_seed(1)
  pts = (torch.rand((M, len(grid)), generator=g, device='cuda') * 2 - 1) * np.pi
  src = torch.complex(torch.rand([M] if ttype == 'type_1' else grid, generator=g, device='cuda'), torch.rand([M] if ttype == 'type_1' else grid, generator=g, device='cuda'))
  plan = tfft.Plan(ttype, grid, 'forward', tol=1e-6)
  plan.set_points(pts)
  for _ in range(3): plan.execute(src)
  plan.set_timing(True); plan.get_timing()
  for _ in range(steps): plan.execute(src)
  tm = plan.get_timing()
  plan.close()
  return tm['fft'][0] / max(tm['fft'][1], 1) * 1e3

cases = [[960, 960], [1000, 1000], [768, 768], [1280, 1280], [1536, 1536], [2048, 2048], [240] * 3, [200] * 3, [320] * 3, [1024, 1024], [256] * 3]
print('grid            type  ' + ' '.join(f'R={r:<6}' for r in ('auto', 16, 8, 4, 2, 1)))
for grid in cases:
  for tt in ('type_1', 'type_2'):
    row = []
    for r in (0, 16, 8, 4, 2, 1):
      if r: os.environ['NUFFT_MIX_R'] = str(r)
      else: os.environ.pop('NUFFT_MIX_R', None)
      try:
        row.append(f'{fft_us(tt, grid):8.1f}')
      except Exception as e:
        row.append('     n/a')
    print(f'{"x".join(map(str, grid)):15} {tt[-1]}    ' + ' '.join(row), flush=True)
