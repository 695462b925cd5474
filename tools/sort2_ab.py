"""A/B of the one-level and the two-level sort of 3-D float plans (options.tuning SORT2_OFF / SORT2_ON):
stage times (plan timing events, median of 4 calls after 2) over point counts on a 512^3 fine grid, both
transform types and tile depths, and config 4's own size. Output: profiles/r03_sort2_ab.txt."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
from tensorflow_nufft import _lib

def run(grid, M, ttype, tol, name, one_call=False):
  g = torch.Generator(device='cuda').manual_seed(1)
  pts = (torch.rand((M, 3), generator=g, device='cuda') * 2 - 1) * np.pi
  shape = M if ttype == 'type_1' else grid
  src = torch.complex(torch.randn(shape, generator=g, device='cuda'), torch.randn(shape, generator=g, device='cuda'))
  plan = tfft.Plan(ttype, grid, 'forward', tol=tol, dtype=torch.complex64, tuning=_lib.TUNE[name] if name else 0)
  plan.set_timing(True)
  acc = []
  for it in range(6):
    if one_call:
      out = plan.execute_with_points(pts, src)
    else:
      plan.set_points(pts); out = plan.execute(src)
    torch.cuda.synchronize()
    t = plan.get_timing()
    if it >= 2: acc.append({k: v[0] / max(1, v[1]) * 1000 for k, v in t.items() if v[1]})
  st = {k: round(float(np.median([a[k] for a in acc]))) for k in acc[0]}
  sort = sum(v for k, v in st.items() if k.startswith('sort'))
  total = sum(st.values())
  print(f'{"x".join(map(str, grid))} M={M:>9} {ttype} tol={tol:g} {"one-call" if one_call else "two-call"} {name or "default":9} '
        f'sort path {plan.sort_path()}: sort {sort:5d} us, all stages {total:6d} us  {st}', flush=True)
  plan.close()

if __name__ == '__main__':
  big = '--big' in sys.argv
  for tol in (1e-4, 1e-6):
    for ttype in ('type_2', 'type_1'):
      for M in (500_000, 1_000_000, 2_000_000, 4_000_000, 10_000_000):
        for name in ('SORT2_OFF', 'SORT2_ON'):
          run([256, 256, 256], M, ttype, tol, name)
  if big:
    for name in ('SORT2_OFF', 'SORT2_ON'):
      run([256, 256, 256], 100_000_000, 'type_2', 1e-4, name)
      run([256, 256, 256], 100_000_000, 'type_1', 1e-4, name)
      run([256, 256, 256], 100_000_000, 'type_1', 1e-4, name, one_call=True)
      run([256, 256, 256], 30_000_000, 'type_1', 1e-6, name)
