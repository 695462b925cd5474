#!/usr/bin/env python3
"""Headline benchmark: BASELINE.json `configs[1]` -- 2D type-1 NUFFT, 1024 x 1024
modes, M = 1e7 random points, tol = 1e-6, fp32, on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch of synthetic input already
resident in HBM: `set_points` (fold + tile sort) followed by `execute`
(spread -> rocFFT -> deconvolve) on a plan that is reused across steps, i.e.
what one `tfft.nufft(source, points, grid_shape, 'type_1')` call costs once its
plan is cached. With N > 1 (launched by torch.distributed.run, one rank per
GPU) every rank runs the same workload on its own points/strengths -- the
transforms of a batch are independent, so the path shards with no data-path
collective ("weak" scaling); RCCL is used only for the barrier and the
max-over-ranks time.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      dominant kernel (spread) against the HBM roofline, from HIP
                events recorded on the plan's stream inside the timed region
  cpu_baseline  the CPU oracle (a port of the reference CPU path, NOT the
                upstream binary) timed on this host's cores, rank 0 at N = 1
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))

import numpy as np   # noqa: E402
import torch         # noqa: E402

GRID = [1024, 1024]
M = 10_000_000
TOL = 1e-6
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_spread_bytes(m, nf, rank):
  # read each point's coordinates + complex strength once,
  # write each fine-grid cell once: M (4 d + 8) + 8 nf^d   [fp32]
  cells = 1
  for n in nf:
    cells *= n
  return m * (4 * rank + 8) + 8 * cells


# dominant kernel of the headline config (2-D, w = 8, float, >= 0.5 points per fine cell)
SPREAD_KERNEL = 'spread_2d_w8_group_kernel'


def pmc_traffic(kernel_name, m):
  """HBM-side bytes per launch of the spread kernel from the PMC passes committed
  under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs,
  corrected as MI355X_MICROARCH.md prescribes: 2*FETCH_SIZE + WRITE_SIZE).
  bench.py cannot run the profiler on itself, so the figure is the one measured
  offline for the same kernel and point count; None when it does not apply."""
  try:
    d = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_spread_traffic.json')))
    if d.get('points') == m and kernel_name in d.get('kernel', ''):
      return int(d['traffic_bytes_corrected'])
  except (OSError, ValueError):
    pass
  return None


def cpu_baseline(args):
  """Times the CPU oracle on a bounded sample of the same workload."""
  from oracle import oracle
  cores = os.cpu_count() or 1
  m = args.cpu_points
  rng = np.random.default_rng(2)
  pts = rng.uniform(-np.pi, np.pi, (m, 2)).astype(np.float32)
  c = (rng.uniform(-.5, .5, m) + 1j * rng.uniform(-.5, .5, m)).astype(np.complex64)
  # reference CPU rule for this config: sigma = 1.25, w = 10,
  # float arithmetic, own piecewise-polynomial kernel (kerevalmeth 1), all cores
  # OpenMP thread counts to try: all hardware threads, the physical cores of a
  # 2-way SMT part, and a NUMA-friendly 64 / 32; the best one is reported.
  cand = sorted({cores, max(1, cores // 2), min(cores, 64), min(cores, 32)}, reverse=True)
  best, best_threads, total, runs = float('inf'), cores, 0.0, 0
  for nt in cand:
    for _ in range(2):
      if total > 30.0:
        break
      t0 = time.perf_counter()
      oracle.nufft(c, pts, GRID, 'type_1', 'forward', tol=TOL, kerevalmeth=1, nthreads=nt)
      dt = time.perf_counter() - t0
      total += dt
      runs += 1
      if dt < best:
        best, best_threads = dt, nt
  cores = best_threads
  sigma, w, _, nf = oracle.query(2, GRID, float(np.float32(TOL)), 'f32')
  return {
      'value': round(m / best / 1e6, 3), 'unit': 'Mpts/s', 'cores': cores, 'kind': 'port',
      'sample': f'full type-1 transform (sort+spread+FFT+deconvolve) of {m} of the {M} points on the '
                f'same 1024x1024 grid, reference CPU rule sigma={sigma} w={w} fine grid {nf[0]}x{nf[1]}, '
                f'fp32, best of {runs} runs over thread counts {cand}: {cores} OpenMP threads, {best:.2f} s',
  }


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=50)
  ap.add_argument('--warmup', type=int, default=5)
  ap.add_argument('--points', type=int, default=M)
  ap.add_argument('--cpu-points', type=int, default=M)
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-extras', action='store_true',
                  help='skip the informational two-stream leg (used under rocprofv3 so that its per-kernel '
                       'average covers single-stream launches only)')
  ap.add_argument('--dist-backend', default='nccl', help='nccl (= RCCL, default) | gloo (testing)')
  ap.add_argument('--device', type=int, default=None, help='force a device index (testing)')
  ap.add_argument('--force-dist', action='store_true', help='initialise torch.distributed even at world size 1 (testing)')
  args = ap.parse_args()

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if not torch.cuda.is_available():
    raise SystemExit('bench.py needs a ROCm GPU (the HIP path has no CPU fallback)')
  if args.device is not None:
    local_rank = args.device
  torch.cuda.set_device(local_rank)
  dev = torch.device('cuda', local_rank)
  dist = None
  if world > 1 or args.force_dist:
    import torch.distributed as dist
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29533')
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('WORLD_SIZE', '1')
    if args.dist_backend == 'nccl':
      dist.init_process_group('nccl', device_id=dev)   # "nccl" is RCCL on ROCm
    else:
      dist.init_process_group(args.dist_backend)

  import tensorflow_nufft as tfft
  m = args.points
  # synthetic inputs: points ~ U(-pi, pi)^2,
  # strengths ~ U(-.5,.5) + i U(-.5,.5); seed 2 (+rank so shards differ)
  g = torch.Generator(device=dev).manual_seed(2 + rank)
  pts = (torch.rand((m, 2), generator=g, device=dev) * 2 - 1) * np.pi
  c = torch.complex(torch.rand(m, generator=g, device=dev) - .5,
                    torch.rand(m, generator=g, device=dev) - .5)
  plan = tfft.Plan('type_1', GRID, 'forward', tol=TOL, dtype=torch.complex64, device=dev)
  info = plan.info()
  out = torch.empty(GRID, dtype=torch.complex64, device=dev)

  def step():
    plan.set_points(pts)
    plan.execute(c, out=out)

  for _ in range(args.warmup):
    step()
  # full per-stage breakdown from an untimed pass (14 events per step cost ~7 %)
  plan.set_timing(1)
  plan.get_timing()
  for _ in range(max(3, args.warmup)):
    step()
  stage_all = plan.get_timing()
  # timed region: HIP events around the dominant (spread) kernel only, recorded
  # on the plan's stream, i.e. the stream the kernel is launched on
  plan.set_timing(2)
  plan.get_timing()
  if dist is not None:
    dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(args.steps):
    step()
  torch.cuda.synchronize()
  if dist is not None:
    dist.barrier()
  elapsed = time.perf_counter() - t0
  stages = plan.get_timing()
  if dist is not None:
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

  # exec-only rate (set_points amortised), same plan
  plan.set_timing(False)
  torch.cuda.synchronize()
  t1 = time.perf_counter()
  for _ in range(args.steps):
    plan.execute(c, out=out)
  torch.cuda.synchronize()
  exec_only = (time.perf_counter() - t1) / args.steps

  # informational: two independent transforms in flight on two streams (the sort is
  # memory-bound, the spread LDS-bound, so they overlap); NOT the headline value
  two_stream = None
  if world == 1 and not args.no_extras:
    s2 = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    lanes = []
    for st in s2:
      with torch.cuda.stream(st):
        lanes.append((tfft.Plan('type_1', GRID, 'forward', tol=TOL, dtype=torch.complex64, device=dev),
                      torch.empty(GRID, dtype=torch.complex64, device=dev)))
    def run2(n):
      for i in range(n):
        with torch.cuda.stream(s2[i % 2]):
          lanes[i % 2][0].set_points(pts)
          lanes[i % 2][0].execute(c, out=lanes[i % 2][1])
    run2(4)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    run2(args.steps)
    torch.cuda.synchronize()
    two_stream = m / ((time.perf_counter() - t2) / args.steps) / 1e6
    for pl, _ in lanes:
      pl.close()

  if rank == 0:
    ms_per_step = elapsed / args.steps * 1e3
    value = world * m / (elapsed / args.steps) / 1e6
    nf = [int(info.fine_dims[1]), int(info.fine_dims[0])]
    spread_ms = stages['spread'][0] / max(stages['spread'][1], 1)
    algo = algorithmic_spread_bytes(m, nf, 2)
    achieved = algo / (spread_ms * 1e-3) / 1e9
    result = {
        'metric': 'non-uniform pts/s, 2D type-1 1024^2 tol=1e-6 (set_points + execute)',
        'value': round(value, 2), 'unit': 'Mpts/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {
            'workload': 'BASELINE configs[1]: 2D type-1, 1024x1024 modes, M=1e7 uniform random points, '
                        'tol=1e-6, complex64; per GPU one independent transform per step',
            'points_per_gpu': m, 'grid': GRID, 'fine_grid': nf, 'kernel_width': int(info.kernel_width),
            'upsampling_factor': info.upsampling_factor, 'spread_method': int(info.spread_method),
            'exec_only_Mpts_s': round(m / exec_only / 1e6, 2),
            'two_streams_Mpts_s': None if two_stream is None else round(two_stream, 2),
            'stage_us': {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in stage_all.items() if v[1]},
        },
        'roofline': {
            'bound': 'hbm', 'kernel': SPREAD_KERNEL, 'achieved': round(achieved, 1),
            'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
            'traffic': pmc_traffic(SPREAD_KERNEL, m), 'algorithmic_bytes': algo,
            'kernel_ms': round(spread_ms, 4),
            'note': 'the kernel is bound by the LDS pipe (per point 3/4 ds_read_b128 + ~0.8 ds_add_f64 after '
                    'grouping points by start cell) plus its per-tile LDS sort, not by HBM: DESIGN.md section 4. '
                    'traffic = offline PMC pass '
                    '(profiles/r01_pmc_traffic.txt); the excess over algorithmic_bytes is the 8-byte strength '
                    'gather through the sort permutation (one 64-B sector per random point)',
        },
    }
    if world == 1 and not args.no_cpu_baseline:
      result['cpu_baseline'] = cpu_baseline(args)
    print(json.dumps(result))
  if dist is not None:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
