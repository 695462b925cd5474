import os, sys
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
g = torch.Generator(device='cuda').manual_seed(2)
for n, M in ((1024, 10_000_000), (1024, 4_000_000)):
  pts = (torch.rand((M, 2), generator=g, device='cuda') * 2 - 1) * np.pi
  f = torch.complex(torch.rand((n, n), generator=g, device='cuda') - .5, torch.rand((n, n), generator=g, device='cuda') - .5)
  for rep in range(2):
    for ms in (0, 8192, 4884, 3256, 2442):
      kw = dict(max_subproblem_size=ms) if ms else {}
      plan = tfft.Plan('type_2', [n, n], 'forward', tol=1e-6, **kw)
      for _ in range(3):
        plan.set_points(pts); plan.execute(f)
      plan.set_timing(True); plan.get_timing()
      for _ in range(10):
        plan.set_points(pts); plan.execute(f)
      tm = plan.get_timing()
      print(f'{n}^2 M={M:.0e} max_sub={ms or "default(16384)"}: interp {tm["interp"][0] / 10 * 1e3:.0f} us', flush=True)
      plan.close()
