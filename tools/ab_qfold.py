"""Same-run A/B of the short coordinate fold of the sort kernels (options.tuning QFOLD_OFF = the general fold):
HIP-event stage times of configs 2, 3 and 4, each variant twice. EXPERIMENTS.md section 10.15.

    python tools/ab_qfold.py [path/to/another/libnufft_hip.so]
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import tensorflow_nufft._lib as L
if len(sys.argv) > 1 and sys.argv[1] != '-':
  L.LIB_PATH = os.path.abspath(sys.argv[1])
import numpy as np, torch
import tensorflow_nufft as tfft
def run(name, ttype, grid, M, tol, tuning=0, one_call=False):
  g = torch.Generator(device='cuda').manual_seed(1)
  r = len(grid)
  pts = (torch.rand((M, r), generator=g, device='cuda') * 2 - 1) * np.pi
  c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
  plan = tfft.Plan(ttype, grid, 'forward', tol=tol, tuning=tuning)
  step = (lambda: plan.execute_with_points(pts, c)) if one_call else (lambda: (plan.set_points(pts), plan.execute(c))[1])
  for _ in range(3):
    out = step()
  plan.set_timing(True); plan.get_timing()
  for _ in range(10):
    out = step()
  tm = plan.get_timing()
  print(name, 'tuning', tuning, ' '.join(f'{k}={v[0] / max(v[1], 1) * 1e3:.1f}us' for k, v in tm.items() if v[1]), 'checksum %.6e' % float(out.abs().sum()), flush=True)
  plan.close()
from tensorflow_nufft._lib import TUNE
for rep in range(2):
  for t in (0, TUNE['QFOLD_OFF']):
    run('cfg2', 'type_1', [1024, 1024], 10_000_000, 1e-6, t)
    run('cfg2 one call', 'type_1', [1024, 1024], 10_000_000, 1e-6, t, one_call=True)
    run('cfg3', 'type_2', [1024, 1024], 10_000_000, 1e-6, t)
    run('cfg4', 'type_1', [256, 256, 256], 100_000_000, 1e-4, t)
