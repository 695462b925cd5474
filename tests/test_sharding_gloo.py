"""Multi-process (world_size 2, gloo, CPU) test of the batch-sharding host
logic used for N > 1 GPUs: contiguous balanced blocks, no data-path collective,
optional all_gather of ragged blocks. The per-item transform is a stand-in
(dense NUDFT in numpy) so that no GPU is needed."""
import os
import socket
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
  s = socket.socket()
  s.bind(('127.0.0.1', 0))
  p = s.getsockname()[1]
  s.close()
  return p


def _worker(rank, world, port, tmpdir):
  for p in (ROOT, PKG):
    if p not in sys.path:
      sys.path.insert(0, p)
  import torch.distributed as dist
  from tensorflow_nufft import sharding
  from oracle import oracle
  os.environ['MASTER_ADDR'] = '127.0.0.1'
  os.environ['MASTER_PORT'] = str(port)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  rng = np.random.default_rng(5)
  B, M, grid = 5, 40, [6, 8]
  pts = rng.uniform(-np.pi, np.pi, (B, M, 2))
  c = rng.standard_normal((B, M)) + 1j * rng.standard_normal((B, M))

  def transform(src, p):
    out = [oracle.nudft(src[i].numpy(), p[i].numpy(), grid, 'type_1', 'forward') for i in range(src.shape[0])]
    return torch.from_numpy(np.stack(out)) if out else torch.zeros((0, 6, 8), dtype=torch.complex128)

  full = sharding.nufft_sharded(torch.from_numpy(c), torch.from_numpy(pts), transform)
  local = sharding.nufft_sharded(torch.from_numpy(c), torch.from_numpy(pts), transform, gather=False)
  lo, hi = sharding.shard_bounds(B, world, rank)
  ref = np.stack([oracle.nudft(c[i], pts[i], grid, 'type_1', 'forward') for i in range(B)])
  ok = (np.allclose(full.numpy(), ref) and np.allclose(local.numpy(), ref[lo:hi]) and
        tuple(full.shape) == (B, 6, 8))
  # shared points variant
  full2 = sharding.nufft_sharded(torch.from_numpy(c), torch.from_numpy(pts[0]),
                                 lambda s, p: torch.from_numpy(np.stack(
                                     [oracle.nudft(s[i].numpy(), p.numpy(), grid, 'type_1', 'forward')
                                      for i in range(s.shape[0])])))
  ref2 = np.stack([oracle.nudft(c[i], pts[0], grid, 'type_1', 'forward') for i in range(B)])
  ok = ok and np.allclose(full2.numpy(), ref2)
  open(os.path.join(tmpdir, f'ok{rank}'), 'w').write('1' if ok else '0')
  dist.destroy_process_group()


def test_shard_bounds():
  from tensorflow_nufft import sharding
  for n in (0, 1, 5, 8, 256, 257):
    for w in (1, 2, 3, 8):
      spans = [sharding.shard_bounds(n, w, r) for r in range(w)]
      assert spans[0][0] == 0 and spans[-1][1] == n
      assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
      sizes = [b - a for a, b in spans]
      assert max(sizes) - min(sizes) <= 1
  assert sharding.shard_bounds(256, 8, 3) == (96, 128)   # BASELINE config 5: 32 items per GPU


def test_world_size_2_gloo(tmp_path):
  port = _free_port()
  mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
  assert open(tmp_path / 'ok0').read() == '1' and open(tmp_path / 'ok1').read() == '1'
