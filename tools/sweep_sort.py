import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
M = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
g = torch.Generator(device='cuda').manual_seed(2)
pts = (torch.rand((M, 2), generator=g, device='cuda') * 2 - 1) * np.pi
plan = tfft.Plan('type_1', [1024, 1024], 'forward', tol=1e-6)
for _ in range(3): plan.set_points(pts)
plan.set_timing(True); plan.get_timing()
for _ in range(10): plan.set_points(pts)
tm = plan.get_timing()
print(os.environ.get('NUFFT_HIP_SORT_BLOCKS', 'default'), M, ' '.join(f"{k}={v[0]/v[1]*1e3:.0f}us" for k, v in tm.items() if v[1]))
