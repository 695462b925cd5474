"""r06: complex128 transforms beside their complex64 twins (same grid, M, tolerance where float reaches it), per-stage
HIP-event times -> profiles/r06_c128.txt. The reference registers complex128 kernels for all three ops
(nufft_kernels.cc:624-706)."""
import os, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch
from bench_nonpow2 import run

if __name__ == '__main__':
  print(torch.cuda.get_device_name(0))
  c64, c128 = torch.complex64, torch.complex128
  which = sys.argv[1:] or ['2d', '3d']
  if '2d' in which:
    for tt in ('type_1', 'type_2'):
      run(tt, [1024, 1024], 10_000_000, 1e-6, 5, dtype=c64)
      run(tt, [1024, 1024], 10_000_000, 1e-6, 5, dtype=c128)
      run(tt, [1024, 1024], 10_000_000, 1e-9, 5, dtype=c128)
      run(tt, [1024, 1024], 10_000_000, 1e-12, 5, dtype=c128)
  if '3d' in which:
    for tt in ('type_1', 'type_2'):
      for M in (10_000_000, 30_000_000):
        run(tt, [256, 256, 256], M, 1e-6, 3, dtype=c64)
        run(tt, [256, 256, 256], M, 1e-6, 3, dtype=c128)
        run(tt, [256, 256, 256], M, 1e-9, 3, dtype=c128)
      run(tt, [128, 128, 128], 20_000_000, 1e-4, 3, dtype=c64)
      run(tt, [128, 128, 128], 20_000_000, 1e-4, 3, dtype=c128)
      run(tt, [128, 128, 128], 10_000_000, 1e-12, 3, dtype=c128)
  if 'tile' in which:   # experiment: deeper tiles for the double-precision 3-D interpolation
    for td in ([16, 16, 4], [16, 16, 8], [16, 8, 8], [8, 8, 8]):
      print('tile_dims', td)
      run('type_2', [256, 256, 256], 10_000_000, 1e-6, 3, dtype=c128, tile_dims=td)
      run('type_2', [256, 256, 256], 30_000_000, 1e-6, 3, dtype=c128, tile_dims=td)
  if 'isplit' in which:
    from tensorflow_nufft._lib import TUNE
    for dt, tols in (((c128, (1e-6, 1e-4)),) if 'c128only' in which else ((c128, (1e-6, 1e-4)), (c64, (1e-6, 1e-4)))):
      for M in (1_000_000, 3_000_000, 10_000_000, 30_000_000, 100_000_000):
        for tol in tols:
          run('type_2', [256, 256, 256], M, tol, 3, dtype=dt, tuning=TUNE['ISPLIT_ON'])
          run('type_2', [256, 256, 256], M, tol, 3, dtype=dt, tuning=TUNE['ISPLIT_OFF'])
  if 'istack' in which:
    from tensorflow_nufft._lib import TUNE
    for M in (3_000_000, 10_000_000, 30_000_000, 100_000_000):
      for tol in (1e-6, 1e-4):
        run('type_2', [256, 256, 256], M, tol, 3, dtype=c128, tuning=TUNE['STACK_ON'])
        run('type_2', [256, 256, 256], M, tol, 3, dtype=c128, tuning=TUNE['STACK_OFF'])
  if 'fistack' in which:   # float 3-D type 2 over stacks (on request only)
    from tensorflow_nufft._lib import TUNE
    for M in (1_000_000, 3_000_000, 10_000_000, 30_000_000, 100_000_000):
      for tol in (1e-6, 1e-4):
        run('type_2', [256, 256, 256], M, tol, 3, dtype=c64, tuning=TUNE['STACK_ON'])
        run('type_2', [256, 256, 256], M, tol, 3, dtype=c64, tuning=TUNE['STACK_OFF'])
  if 'wideistack' in which:
    from tensorflow_nufft._lib import TUNE
    for M in (3_000_000, 10_000_000, 30_000_000):
      for grid, tol in (([256] * 3, 1e-9), ([128] * 3, 1e-12)):
        run('type_2', grid, M, tol, 3, dtype=c128, tuning=TUNE['STACK_ON'])
        run('type_2', grid, M, tol, 3, dtype=c128, tuning=TUNE['STACK_OFF'])
  if 'widestack' in which:
    from tensorflow_nufft._lib import TUNE
    for M in (10_000_000, 30_000_000):
      for grid, tol in (([256] * 3, 1e-9), ([128] * 3, 1e-12)):
        run('type_1', grid, M, tol, 3, dtype=c128, tuning=TUNE['STACK_ON'])
        run('type_1', grid, M, tol, 3, dtype=c128, tuning=TUNE['STACK_OFF'])
  if 'stack' in which:
    from tensorflow_nufft._lib import TUNE
    for M in (3_000_000, 10_000_000, 30_000_000, 100_000_000, 300_000_000):
      for tol in (1e-6, 1e-4):
        run('type_1', [256, 256, 256], M, tol, 3, dtype=c128, tuning=TUNE['STACK_ON'])
        run('type_1', [256, 256, 256], M, tol, 3, dtype=c128, tuning=TUNE['STACK_OFF'])
