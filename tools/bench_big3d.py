"""r06: 3-D fine grids past the two-level sort's 1024 super-tiles (fine 768^3, 1024^3): stage times, which sort ran."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np, torch, time
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import tensorflow_nufft as tfft
print(torch.cuda.get_device_name(0))
for N, M, tol in ((320, 30_000_000, 1e-6), (384, 30_000_000, 1e-6), (384, 100_000_000, 1e-4), (512, 30_000_000, 1e-6), (512, 100_000_000, 1e-4)):
  for tt in ('type_1', 'type_2'):
    g = torch.Generator(device='cuda').manual_seed(3)
    pts = (torch.rand((M, 3), generator=g, device='cuda') * 2 - 1) * np.pi
    src = torch.complex(torch.rand([M] if tt == 'type_1' else [N] * 3, generator=g, device='cuda') - .5, torch.rand([M] if tt == 'type_1' else [N] * 3, generator=g, device='cuda') - .5)
    plan = tfft.Plan(tt, [N] * 3, 'forward', tol=tol)
    def step():
      plan.set_points(pts); plan.execute(src)
    step(); step()
    plan.set_timing(True); plan.get_timing()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    tm = plan.get_timing()
    st = {k: v[0] / max(v[1], 1) * 1e3 for k, v in tm.items() if v[1]}
    print(f'{tt[-1]} {N}^3 M={M:.0e} tol={tol:g} sort_path={plan.sort_path()}: {dt*1e3:8.2f} ms/step | ' + ' '.join(f'{k}={v:.0f}' for k, v in st.items()), flush=True)
    plan.close(); del pts, src; torch.cuda.empty_cache()
