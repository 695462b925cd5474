"""3-D non-uniform point sets at 256^3 modes, fp32: uniform against a radial ("kooshball": density ~ 1/r^2) trajectory in
random and in acquisition order, at the config-4 tolerance (1e-4, w = 6: spread_dense3_kernel) and at the default
tolerance (1e-6, w = 8: spread_patch3 / spread_stack3_kernel). Per case: ms per one-call transform, HIP-event stage times,
how many subproblems / stacks the count-filter bound sent to the fp64 planes, and (type 1, --acc) the error against a
double-precision tol 1e-9 transform of the same data.

    python tools/exp_clustered3d.py [M] [--tols 1e-4,1e-6] [--acc]
"""
import argparse, os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.environ.get('NUFFT_PKG', os.path.join(ROOT, 'tensorflow-nufft_amd')))
import numpy as np, torch
import tensorflow_nufft as tfft
ap = argparse.ArgumentParser()
ap.add_argument('M', nargs='?', type=float, default=3e7)
ap.add_argument('--tols', default='1e-4,1e-6')
ap.add_argument('--acc', action='store_true')
ap.add_argument('--grid', type=int, default=256)
args = ap.parse_args()
M = int(args.M)
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
grid = [args.grid] * 3
print(f'# 3-D {args.grid}^3 modes, M = {M:.3g} ({M / (2 * args.grid) ** 3:.3f} per fine cell), complex64; ms per one-call transform (tfft Plan.execute_with_points), stage times in us')
for tol in [float(t) for t in args.tols.split(',')]:
  for name, pts in cases.items():
    m = pts.shape[0]
    c = torch.complex(torch.rand(m, generator=g, device='cuda') - .5, torch.rand(m, generator=g, device='cuda') - .5)
    f = torch.complex(torch.rand(grid, generator=g, device='cuda') - .5, torch.rand(grid, generator=g, device='cuda') - .5)
    for tt, src in (('type_1', c), ('type_2', f)):
      plan = tfft.Plan(tt, grid, 'forward', tol=tol)
      out = plan.execute_with_points(pts, src)
      plan.set_timing(True); plan.get_timing()
      torch.cuda.synchronize(); t0 = time.perf_counter()
      for _ in range(3): out = plan.execute_with_points(pts, src)
      torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
      tm = plan.get_timing()
      line = f'tol {tol:g} w={plan.info().kernel_width} {name:15s} {tt}: {dt*1e3:8.3f} ms ({m / dt / 1e9:.2f} Gpts/s) ' + ' '.join(f'{k}={v[0] / v[1] * 1e3:.0f}' for k, v in tm.items() if v[1])
      if tt == 'type_1':
        plan.set_points(pts)
        b = plan.sub_bounds()
        live = b[b != 0]
        if live.size:
          st = plan.stacks()
          line += f' | {"stacks" if st.size else "subproblems"} {live.size}, bound mean {np.abs(live).mean():.1f} max {np.abs(live).max():.0f}, on fp64 planes {int((live < 0).sum())}'
          if st.size:
            line += f' (pieces {int((st[:, 2] >= 0).sum())})'
        if args.acc:
          ref = tfft.nufft(src.to(torch.complex128), pts.double(), grid_shape=grid, transform_type='type_1', tol=1e-9)
          line += f' | rel-l2 vs fp64 tol 1e-9: {float(torch.linalg.norm(out.to(torch.complex128) - ref) / torch.linalg.norm(ref)):.3e}'
          del ref
      print(line, flush=True)
      plan.close()
      torch.cuda.empty_cache()
