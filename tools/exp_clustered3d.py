"""3-D non-uniform point sets at the config-4 geometry (256^3, tol 1e-4, fp32): uniform against a radial
("kooshball": density ~ 1/r^2) trajectory in random and in acquisition order; ms per one-call transform."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
M = int(float(sys.argv[1])) if len(sys.argv) > 1 else 30_000_000
g = torch.Generator(device='cuda').manual_seed(4)
def radial(n, ordered):
  ns = 500; nsp = n // ns
  u = torch.rand(nsp, generator=g, device='cuda') * 2 - 1; ph = torch.rand(nsp, generator=g, device='cuda') * 2 * np.pi
  d = torch.stack([torch.sqrt(1 - u * u) * torch.cos(ph), torch.sqrt(1 - u * u) * torch.sin(ph), u], dim=1)   # spoke directions
  s = torch.linspace(-np.pi, np.pi, ns + 1, device='cuda')[:ns]
  p = (d[:, None, :] * s[None, :, None]).reshape(-1, 3)
  return p if ordered else p[torch.randperm(p.shape[0], device='cuda', generator=g)]
cases = {'uniform': (torch.rand((M, 3), generator=g, device='cuda') * 2 - 1) * np.pi,
         'radial': radial(M, False), 'radial-ordered': radial(M, True)}
grid = [256, 256, 256]
for name, pts in cases.items():
  m = pts.shape[0]
  c = torch.complex(torch.rand(m, generator=g, device='cuda') - .5, torch.rand(m, generator=g, device='cuda') - .5)
  f = torch.complex(torch.rand(grid, generator=g, device='cuda') - .5, torch.rand(grid, generator=g, device='cuda') - .5)
  for tt, src in (('type_1', c), ('type_2', f)):
    plan = tfft.Plan(tt, grid, 'forward', tol=1e-4)
    out = plan.execute_with_points(pts, src)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): out = plan.execute_with_points(pts, src)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f'{name:15s} {tt}: {dt*1e3:8.3f} ms  ({m / dt / 1e9:.2f} Gpts/s)')
    plan.close()
