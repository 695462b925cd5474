import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
M = 10_000_000
g = torch.Generator(device='cuda').manual_seed(2)
pts = (torch.rand((M, 2), generator=g, device='cuda') * 2 - 1) * np.pi
# tile id of each point (approximate: by position) and a tile-sorted copy
t = ((pts + np.pi) * (2048 / (2 * np.pi))).floor().long().clamp(0, 2047) // 32
key = t[:, 0] * 64 + t[:, 1]     # points[:,1] is x (fastest)
order = torch.argsort(key)
pts_sorted = pts[order].contiguous()
c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
f = torch.complex(torch.rand((1024, 1024), generator=g, device='cuda') - .5, torch.rand((1024, 1024), generator=g, device='cuda') - .5)
for name, p in (('random', pts), ('tile-sorted', pts_sorted)):
  for tt, src in (('type_1', c), ('type_2', f)):
    plan = tfft.Plan(tt, [1024, 1024], 'forward', tol=1e-6)
    plan.set_points(p); out = plan.execute(src)
    for _ in range(3): plan.set_points(p); plan.execute(src, out=out)
    plan.set_timing(True); plan.get_timing()
    for _ in range(10): plan.set_points(p); plan.execute(src, out=out)
    tm = plan.get_timing()
    print(name, tt, ' '.join(f"{k}={v[0]/v[1]*1e3:.0f}" for k, v in tm.items() if v[1]))
    plan.close()
