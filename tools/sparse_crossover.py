"""Crossover between the LDS-tile spreaders and the LDS-free (global atomic) spreader:
spread-stage time by point count, both forced (options.tuning SPARSE_OFF / SPARSE_ON; each
measurement in a child process)."""
import os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
CHILD = r'''
import sys, os
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft
grid = [int(g) for g in sys.argv[2].split(',')]; M = int(float(sys.argv[3])); tol = float(sys.argv[4])
g = torch.Generator(device='cuda').manual_seed(1)
pts = (torch.rand((M, len(grid)), generator=g, device='cuda') * 2 - 1) * np.pi
c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
from tensorflow_nufft import _lib
plan = tfft.Plan('type_1', grid, tol=tol, tuning=_lib.TUNE[sys.argv[5]])
plan.set_points(pts)
for _ in range(3): plan.execute(c)
plan.set_timing(2); plan.get_timing()
for _ in range(10): plan.execute(c)
t = plan.get_timing()['spread']
print(t[0] / t[1] * 1e3)
'''
for grid, tol, Ms in (('256,256,256', 1e-4, ['1e3', '1e4', '1e5', '3e5', '1e6', '3e6', '1e7', '3e7']),
                      ('1024,1024', 1e-6, ['1e2', '1e3', '1e4', '3e4', '1e5', '3e5', '1e6'])):
  print(f'# grid {grid} tol {tol}: spread stage, us   (points | points per fine cell | LDS-tile kernels | LDS-free kernel)')
  cells = 1
  for gdim in grid.split(','): cells *= 2 * int(gdim)
  for M in Ms:
    res = []
    for mode in ('SPARSE_OFF', 'SPARSE_ON'):
      r = subprocess.run([sys.executable, '-c', CHILD, ROOT, grid, M, str(tol), mode], capture_output=True, text=True)
      res.append(float(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else float('nan'))
    print(f'{M:>6} | {float(M)/cells:9.2e} | {res[0]:10.1f} | {res[1]:10.1f}')
