#!/usr/bin/env python3
"""HIP-event stage times of one transform: python tools/stage_times.py type_2 128,128,128 8e5 [tol] [tuning,...] [--one-call]"""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.environ.get('NUFFT_PKG', os.path.join(ROOT, 'tensorflow-nufft_amd')))
import numpy as np, torch
import tensorflow_nufft as tfft
from tensorflow_nufft import _lib
one_call = '--one-call' in sys.argv
argv = [a for a in sys.argv[1:] if a != '--one-call']
ttype, grid, M = argv[0], [int(g) for g in argv[1].split(',')], int(float(argv[2]))
tol = float(argv[3]) if len(argv) > 3 else 1e-6
tuning = 0
for b in filter(None, (argv[4] if len(argv) > 4 else '').split(',')): tuning |= _lib.TUNE[b]
g = torch.Generator(device='cuda').manual_seed(2)
pts = (torch.rand((M, len(grid)), generator=g, device='cuda') * 2 - 1) * np.pi
shape = [M] if ttype == 'type_1' else grid
src = torch.complex(torch.rand(shape, generator=g, device='cuda') - .5, torch.rand(shape, generator=g, device='cuda') - .5)
plan = tfft.Plan(ttype, grid, 'forward', tol=tol, tuning=tuning)
def step():
  if one_call: return plan.execute_with_points(pts, src)
  plan.set_points(pts); return plan.execute(src)
for _ in range(3): step()
plan.set_timing(True); plan.get_timing()
for _ in range(10): step()
tm = plan.get_timing()
i = plan.info()
st = {k: v[0] / 10 * 1e3 for k, v in tm.items() if v[1]}   # (per call)
print(f'{ttype} {grid} M={M:.3g} tol={tol:g} w={i.kernel_width} tile={list(i.tile_dims)}: ' + ' '.join(f'{k}={v:.0f}us' for k, v in st.items()) + f' | all {sum(st.values()) / 1e3:.3f} ms')
