"""Exploratory GPU probe: accuracy + stage timings for the headline config."""
import os, sys, time
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft

def timeit(fn, n=5, warm=2):
  for _ in range(warm): fn()
  torch.cuda.synchronize()
  t = time.perf_counter()
  for _ in range(n): fn()
  torch.cuda.synchronize()
  return (time.perf_counter() - t) / n

def main():
  print(torch.cuda.get_device_name(0))
  M = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
  g = torch.Generator(device='cuda').manual_seed(2)
  pts = (torch.rand((M, 2), generator=g, device='cuda') * 2 - 1) * np.pi
  c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
  for method in (2, 1):
    for S in (1024, 4096):
      plan = tfft.Plan('type_1', [1024, 1024], 'forward', tol=1e-6, spread_method=method, max_subproblem_size=S)
      i = plan.info()
      t_set = timeit(lambda: plan.set_points(pts))
      out = plan.execute(c)
      t_exec = timeit(lambda: plan.execute(c, out=out))
      plan.set_timing(True)
      for _ in range(5):
        plan.set_points(pts); plan.execute(c, out=out)
      tm = plan.get_timing()
      print('   stages(us):', ' '.join(f'{k}={v[0]/max(v[1],1)*1e3:.0f}' for k, v in tm.items() if v[1]))
      print(f'method {method} S {S} w {i.kernel_width} nc {i.ncoef}: set_points {t_set*1e3:.3f} ms  execute {t_exec*1e3:.3f} ms  '
            f'-> {M/(t_set+t_exec)/1e6:.1f} Mpts/s total, {M/t_exec/1e6:.1f} Mpts/s exec')
      plan.close()
  plan2 = tfft.Plan('type_2', [1024, 1024], 'forward', tol=1e-6)
  f = torch.complex(torch.rand((1024, 1024), generator=g, device='cuda') - .5, torch.rand((1024, 1024), generator=g, device='cuda') - .5)
  plan2.set_points(pts)
  o2 = plan2.execute(f)
  t2 = timeit(lambda: plan2.execute(f, out=o2))
  print(f'type 2 execute {t2*1e3:.3f} ms -> {M/t2/1e6:.1f} Mpts/s')

if __name__ == '__main__':
  main()
