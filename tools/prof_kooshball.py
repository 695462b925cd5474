# rocprofv3 target: the 256^3 kooshball at the default tolerance, M = 3e7, four one-call type-1 transforms (profiles/r05_kooshball_kernels*.txt):
#   rocprofv3 --kernel-trace --stats -d gpurun_out/prof_koosh -o run --output-format csv -- python3 tools/prof_kooshball.py
import os, sys
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
M = 30_000_000
g = torch.Generator(device='cuda').manual_seed(4)
ns = 500; nsp = M // ns
u = torch.rand(nsp, generator=g, device='cuda') * 2 - 1; ph = torch.rand(nsp, generator=g, device='cuda') * 2 * np.pi
d = torch.stack([torch.sqrt(1 - u * u) * torch.cos(ph), torch.sqrt(1 - u * u) * torch.sin(ph), u], dim=1)
s = torch.linspace(-np.pi, np.pi, ns + 1, device='cuda')[:ns]
p = (d[:, None, :] * s[None, :, None]).reshape(-1, 3)
p = p[torch.randperm(p.shape[0], device='cuda', generator=g)]
c = torch.complex(torch.rand(p.shape[0], generator=g, device='cuda') - .5, torch.rand(p.shape[0], generator=g, device='cuda') - .5)
plan = tfft.Plan('type_1', [256] * 3, 'forward', tol=1e-6)
for _ in range(4):
  out = plan.execute_with_points(p, c)
torch.cuda.synchronize()
plan.set_points(p)
b = plan.sub_bounds(); live = b[b != 0]
print('subproblems', live.size, 'flagged', int((live < 0).sum()))
