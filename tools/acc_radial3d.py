import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
g = torch.Generator(device='cuda').manual_seed(4)
M, grid = 30_000_000, [256, 256, 256]
ns = 500; nsp = M // ns
u = torch.rand(nsp, generator=g, device='cuda') * 2 - 1; ph = torch.rand(nsp, generator=g, device='cuda') * 2 * np.pi
d = torch.stack([torch.sqrt(1 - u * u) * torch.cos(ph), torch.sqrt(1 - u * u) * torch.sin(ph), u], dim=1)
s = torch.linspace(-np.pi, np.pi, ns + 1, device='cuda')[:ns]
pts = (d[:, None, :] * s[None, :, None]).reshape(-1, 3)
c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
truth = tfft.nufft(c.to(torch.complex128), pts.double(), grid_shape=grid, transform_type='type_1', tol=1e-9)
for tol in (1e-4, 1e-5):
  out = tfft.nufft(c, pts, grid_shape=grid, transform_type='type_1', tol=tol)
  print(f'3-D radial M=3e7 256^3 float tol {tol:g}: rel-l2 vs fp64 tol 1e-9 transform = {float(torch.linalg.norm(out.to(torch.complex128) - truth) / torch.linalg.norm(truth)):.3e}')
