import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
from tensorflow_nufft import _lib
def relerr(a, b):
  a = a.to(torch.complex128); b = b.to(torch.complex128)
  return float(torch.linalg.norm(a - b) / torch.linalg.norm(b))
def case(grid, M, ttype, tol, one_call, seed=1, clustered=False):
  g = torch.Generator(device='cuda').manual_seed(seed)
  pts = (torch.rand((M, 3), generator=g, device='cuda') * 2 - 1) * np.pi
  if clustered: pts = pts * 0.05 + 1.0
  if ttype == 'type_1':
    src = torch.complex(torch.randn(M, generator=g, device='cuda'), torch.randn(M, generator=g, device='cuda'))
  else:
    src = torch.complex(torch.randn(grid, generator=g, device='cuda'), torch.randn(grid, generator=g, device='cuda'))
  outs = {}
  for name in ('SORT2_OFF', 'SORT2_ON'):
    plan = tfft.Plan(ttype, grid, 'forward', tol=tol, dtype=torch.complex64, tuning=_lib.TUNE[name])
    plan.set_timing(True)
    for _ in range(3):
      if one_call and ttype == 'type_1': out = plan.execute_with_points(pts, src)
      else:
        plan.set_points(pts); out = plan.execute(src)
      torch.cuda.synchronize()
      t = plan.get_timing()
    outs[name] = out
    st = {k: round(v[0] / max(1, v[1]) * 1000) for k, v in t.items() if v[1]}
    print('   ', name, st, flush=True)
  e = relerr(outs['SORT2_ON'], outs['SORT2_OFF'])
  print(grid, M, ttype, tol, 'one_call' if one_call else 'two_call', 'clustered' if clustered else '', 'on-vs-off rel-l2 %.3e' % e, flush=True)
  return e
worst = 0
for args in (([128, 128, 128], 300_000, 'type_1', 1e-4, False), ([128, 128, 128], 300_000, 'type_1', 1e-4, True),
             ([128, 128, 128], 300_000, 'type_2', 1e-4, False), ([64, 64, 96], 100_000, 'type_1', 1e-6, True),
             ([64, 64, 96], 5_000_00, 'type_2', 1e-5, False), ([128, 128, 128], 3_000_000, 'type_1', 1e-4, True),
             ([128, 128, 128], 3_000_000, 'type_2', 1e-4, False)):
  worst = max(worst, case(*args))
worst = max(worst, case([128, 128, 128], 2_000_000, 'type_1', 1e-4, True, clustered=True))
worst = max(worst, case([128, 128, 128], 2_000_000, 'type_2', 1e-4, False, clustered=True))
print('worst', worst)
if len(sys.argv) > 1:
  for args in (([256, 256, 256], 100_000_000, 'type_1', 1e-4, True), ([256, 256, 256], 100_000_000, 'type_2', 1e-4, False),
               ([256, 256, 256], 30_000_000, 'type_1', 1e-6, True)):
    case(*args)
