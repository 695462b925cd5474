"""Type-2 execute on a reused plan: does the cell-sorted record order speed up the LDS gather?
(Needs maybe_cellsort enabled for type 2 in nufft_plan.cpp; r01 answer: bank-conflict cycles
8.2e7 -> 2.2e6 and LDS activity / 2.8 (tools/pmc_interp2.sh), kernel time unchanged at 281 us:
the per-thread dependency chains, not LDS throughput, bound interp_point_kernel.)"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
M, N = 10_000_000, 1024
g = torch.Generator(device='cuda').manual_seed(2)
pts = (torch.rand((M, 2), generator=g, device='cuda') * 2 - 1) * np.pi
f = torch.complex(torch.rand((N, N), generator=g, device='cuda') - .5, torch.rand((N, N), generator=g, device='cuda') - .5)
plan = tfft.Plan('type_2', [N, N], 'forward', tol=1e-6)
plan.set_points(pts)
plan.set_timing(True)
out = plan.execute(f)
ref = out.clone()
for rnd in range(3):
  plan.get_timing()
  for _ in range(4): plan.execute(f, out=out)
  tm = plan.get_timing()
  print(f"round {rnd}: interp {tm['interp'][0]/tm['interp'][1]*1e3:.0f} us, sort_cell calls {tm['sort_cell'][1]}, max diff vs first {float((out-ref).abs().max()):.2e}")
