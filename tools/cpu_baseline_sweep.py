#!/usr/bin/env python3
"""CPU baseline (the oracle = port of the reference CPU path) over OpenMP thread counts and bindings, each in its own
process (the OpenMP runtime reads its environment once): full type-1 transform of config 2, median of 3 after a warm-up."""
import os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
CHILD = r'''
import os, sys, time
sys.path.insert(0, sys.argv[1])
import numpy as np
from oracle import oracle
nt = int(sys.argv[2])
m = 10_000_000
rng = np.random.default_rng(2)
pts = rng.uniform(-np.pi, np.pi, (m, 2)).astype(np.float32)
c = (rng.uniform(-.5, .5, m) + 1j * rng.uniform(-.5, .5, m)).astype(np.complex64)
oracle.nufft(c, pts, [1024, 1024], 'type_1', 'forward', tol=1e-6, kerevalmeth=1, nthreads=nt)
ts = []
for _ in range(3):
  t0 = time.perf_counter()
  oracle.nufft(c, pts, [1024, 1024], 'type_1', 'forward', tol=1e-6, kerevalmeth=1, nthreads=nt)
  ts.append(time.perf_counter() - t0)
sp = oracle.time_spread(c, pts, [1024, 1024], tol=float(np.float32(1e-6)), sigma=0.0, kerevalmeth=1, nthreads=nt)
print(f'{m / sorted(ts)[1] / 1e6:.1f} {m / sp / 1e6:.1f}')
'''
print('# threads | binding | full transform Mpts/s | spreader alone Mpts/s   (host:', os.cpu_count(), 'hardware threads)')
for bind in ('', 'close', 'spread'):
  for nt in (16, 32, 48, 64, 96, 128, 256):
    if nt > (os.cpu_count() or 1):
      continue
    env = dict(os.environ)
    if bind:
      env['OMP_PROC_BIND'] = bind
      env['OMP_PLACES'] = 'cores'
    r = subprocess.run([sys.executable, '-c', CHILD, ROOT, str(nt)], env=env, capture_output=True, text=True, timeout=600)
    print(nt, bind or 'unbound', r.stdout.strip() or r.stderr[-200:], flush=True)
