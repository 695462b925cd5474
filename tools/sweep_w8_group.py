"""A/B of the cell-grouped w = 8 spread kernel (NUFFT_HIP_W8_GROUP, NUFFT_HIP_W8_SHAPE are
read once per process, so each setting runs in a child process)."""
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
  plan = tfft.Plan('type_1', [N, N], 'forward', tol=1e-6)
  plan.set_points(pts); out = plan.execute(c)
  for _ in range(3): plan.execute(c, out=out)
  plan.set_timing(True); plan.get_timing()
  for _ in range(10): plan.execute(c, out=out)
  tm = plan.get_timing()
  res.append(f"M={M:.0e},N={N}: {tm['spread'][0]/tm['spread'][1]*1e3:.0f}us s={out.abs().double().sum().item():.7e}")
  plan.close()
print(os.environ.get('NUFFT_HIP_W8_GROUP', 'auto'), os.environ.get('NUFFT_HIP_W8_SHAPE', 'default'), ' ; '.join(res))
''' % (ROOT, ROOT)
for grp in ('0', '1'):
  for shape in (['4x64'] if grp == '0' else ['8x64', '12x64', '12x32']):
    env = dict(os.environ, NUFFT_HIP_W8_GROUP=grp, NUFFT_HIP_W8_SHAPE=shape)
    r = subprocess.run([sys.executable, '-c', CHILD], env=env, capture_output=True, text=True)
    print(r.stdout.strip() or r.stderr[-800:], flush=True)
