"""Two plans on two streams, alternating independent transforms: does the
memory-bound sort of one overlap the LDS-bound spread of the other?"""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
M = int(float(os.environ.get('EXP_M', '1e7')))
GRID = [int(os.environ.get('EXP_N', '1024'))] * 2
g = torch.Generator(device='cuda').manual_seed(2)
NP = int(sys.argv[1]) if len(sys.argv) > 1 else 2
pts = [(torch.rand((M, 2), generator=g, device='cuda') * 2 - 1) * np.pi for _ in range(NP)]
cs = [torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5) for _ in range(NP)]
streams = [torch.cuda.Stream() for _ in range(NP)]
plans, outs = [], []
for i in range(NP):
  with torch.cuda.stream(streams[i]):
    plans.append(tfft.Plan('type_1', GRID, 'forward', tol=1e-6))
    outs.append(torch.empty(GRID, dtype=torch.complex64, device='cuda'))
def run(steps):
  for s in range(steps):
    i = s % NP
    with torch.cuda.stream(streams[i]):
      plans[i].set_points(pts[i]); plans[i].execute(cs[i], out=outs[i])
run(2 * NP); torch.cuda.synchronize()
K = 60
t0 = time.perf_counter(); run(K); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(f'{NP} plans/streams: {dt*1e3:.3f} ms per transform -> {M/dt/1e6:.0f} Mpts/s')
