"""A/B of the cell-grouped against the per-point w = 8 spread kernel (options.tuning GROUP_ON / GROUP_OFF)."""
import os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
CHILD = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
res = []
ref = None
for M, N in ((10_000_000, 1024), (1_000_000, 512), (4_000_000, 1024)):
  g = torch.Generator(device='cuda').manual_seed(2)
  pts = (torch.rand((M, 2), generator=g, device='cuda') * 2 - 1) * np.pi
  c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
  from tensorflow_nufft import _lib
  plan = tfft.Plan('type_1', [N, N], 'forward', tol=1e-6, tuning=_lib.TUNE[sys.argv[1]])
  plan.set_points(pts); out = plan.execute(c)
  for _ in range(3): plan.execute(c, out=out)
  plan.set_timing(True); plan.get_timing()
  for _ in range(10): plan.execute(c, out=out)
  tm = plan.get_timing()
  res.append(f"M={M:.0e},N={N}: {tm['spread'][0]/tm['spread'][1]*1e3:.0f}us s={out.abs().double().sum().item():.7e}")
  plan.close()
print(sys.argv[1], ' ; '.join(res))
''' % (ROOT, ROOT)
for grp in ('GROUP_OFF', 'GROUP_ON'):
  r = subprocess.run([sys.executable, '-c', CHILD, grp], capture_output=True, text=True)
  print(r.stdout.strip() or r.stderr[-800:], flush=True)
