"""Does set_points + execute capture into a HIP graph (torch.cuda.CUDAGraph) and what does replay cost?"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
M = int(float(os.environ.get('EXP_M', '1e6'))); N = int(os.environ.get('EXP_N', '512'))
g = torch.Generator(device='cuda').manual_seed(2)
pts = (torch.rand((M, 2), generator=g, device='cuda') * 2 - 1) * np.pi
c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
  plan = tfft.Plan('type_1', [N, N], 'forward', tol=1e-6)
  out = torch.empty((N, N), dtype=torch.complex64, device='cuda')
  for _ in range(3):
    plan.set_points(pts); plan.execute(c, out=out)
  s.synchronize()
  ref = out.clone()
  t0 = time.perf_counter()
  for _ in range(200):
    plan.set_points(pts); plan.execute(c, out=out)
  s.synchronize(); eager = (time.perf_counter() - t0) / 200
  graph = torch.cuda.CUDAGraph()
  out.zero_()
  with torch.cuda.graph(graph, stream=s):
    plan.set_points(pts); plan.execute(c, out=out)
  graph.replay(); s.synchronize()
  err = float((out - ref).abs().max() / ref.abs().max())
  t0 = time.perf_counter()
  for _ in range(200):
    graph.replay()
  s.synchronize(); rep = (time.perf_counter() - t0) / 200
print(f'M={M} N={N}: eager {eager*1e6:.1f} us, graph replay {rep*1e6:.1f} us, max rel diff {err:.2e}')
