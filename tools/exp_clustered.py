"""Non-uniform point distributions: radial (MRI-like, density ~ 1/r) in random and in acquisition
(spoke by spoke) order, gaussian cluster, all-identical. SURVEY.md 8(d): secondary stress inputs."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
M = 10_000_000
g = torch.Generator(device='cuda').manual_seed(2)
r = torch.rand(M, generator=g, device='cuda') * np.pi
th = torch.rand(M, generator=g, device='cuda') * 2 * np.pi
cases = {
  'uniform': (torch.rand((M, 2), generator=g, device='cuda') * 2 - 1) * np.pi,
  'radial': torch.stack([r * torch.cos(th), r * torch.sin(th)], dim=1),
  # 10000 spokes of 1000 samples through the centre, stored spoke after spoke (acquisition order)
  'radial-ordered': (lambda a, s: torch.stack([(s[None, :] * torch.cos(a)[:, None]).reshape(-1), (s[None, :] * torch.sin(a)[:, None]).reshape(-1)], dim=1))(
      torch.arange(10000, device='cuda') * (np.pi * 0.6180339887), torch.linspace(-np.pi, np.pi, 1001, device='cuda')[:1000]),
  'gauss(0.1)': (0.1 * torch.randn((M, 2), generator=g, device='cuda')).clamp(-3, 3),
  'identical': torch.full((M, 2), 0.3, device='cuda'),
}
c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
f = torch.complex(torch.rand((1024, 1024), generator=g, device='cuda') - .5, torch.rand((1024, 1024), generator=g, device='cuda') - .5)
only = os.environ.get('ONLY_TYPE')
for name, pts in cases.items():
  for tt, src in (('type_1', c), ('type_2', f)):
    if only and tt != only: continue
    plan = tfft.Plan(tt, [1024, 1024], 'forward', tol=1e-6)
    for _ in range(2): out = plan.execute_with_points(pts, src)   # (the timed call's own path: first calls allocate)
    plan.set_timing(1); plan.get_timing()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): out = plan.execute_with_points(pts, src)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    tm = plan.get_timing()
    print(f'{name:15s} {tt}: {dt*1e3:8.3f} ms ', ' '.join(f"{k}={v[0]/v[1]*1e3:.0f}" for k, v in tm.items() if v[1]))
    plan.close()
