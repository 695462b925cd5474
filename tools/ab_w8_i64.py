# 3-D float w = 8: fp64 planes (default) against 64-bit integer planes (NUFFT_HIP_W8_I64=1): error vs an fp64 tol 1e-12 transform, and time
import os, sys, time
ROOT = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
M = int(float(sys.argv[1])); n = int(sys.argv[2])
g = torch.Generator(device='cuda').manual_seed(4)
pts = (torch.rand((M, 3), generator=g, device='cuda') * 2 - 1) * np.pi
c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
if len(sys.argv) > 3 and sys.argv[3] == 'spiky': c[::1000] *= 1000.0   # a few huge strengths among small ones
out = tfft.nufft(c, pts, grid_shape=[n] * 3, transform_type='type_1', tol=1e-6)
truth = tfft.nufft(c.to(torch.complex128), pts.double(), grid_shape=[n] * 3, transform_type='type_1', tol=1e-12)
err = float(torch.linalg.norm(out.to(torch.complex128) - truth) / torch.linalg.norm(truth))
for _ in range(2): tfft.nufft(c, pts, grid_shape=[n] * 3, transform_type='type_1', tol=1e-6)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): tfft.nufft(c, pts, grid_shape=[n] * 3, transform_type='type_1', tol=1e-6)
torch.cuda.synchronize()
print(f'M={M:.0e} n={n} {"I64" if os.environ.get("NUFFT_HIP_W8_I64") else "f64"}: rel-l2 {err:.3e}  {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms')
