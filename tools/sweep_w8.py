import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
M = 10_000_000
g = torch.Generator(device='cuda').manual_seed(2)
pts = (torch.rand((M, 2), generator=g, device='cuda') * 2 - 1) * np.pi
c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
res = []
for S in (768, 1024, 1280, 2560, 4096):
  plan = tfft.Plan('type_1', [1024, 1024], 'forward', tol=1e-6, max_subproblem_size=S)
  plan.set_points(pts); out = plan.execute(c)
  for _ in range(3): plan.execute(c, out=out)
  plan.set_timing(True); plan.get_timing()
  for _ in range(10): plan.execute(c, out=out)
  tm = plan.get_timing()
  res.append(f"S={S}:{tm['spread'][0]/tm['spread'][1]*1e3:.0f}us")
  plan.close()
print(os.environ.get('NUFFT_HIP_W8_SHAPE', 'default'), ' '.join(res))
