#!/usr/bin/env python3
"""Benchmark of the NUFFT hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Every N: BASELINE.json `configs[1]`, the configuration the metric is quoted on
  -- 2D type-1 NUFFT, 1024 x 1024 modes, M = 1e7 random points, tol = 1e-6,
  complex64 -- one independent transform per rank and step (transforms are the
  unit the path shards by; a single transform is not split: "weak" scaling,
  per-GPU work fixed, `value` = N x M / max-over-ranks time, no data-path
  collective). A "step" is one pass of the hot path over one batch of synthetic
  input already resident in HBM: what one
  `tfft.nufft(source, points, grid_shape, 'type_1')` call runs once its plan is
  cached, i.e. `nufft_hip_execute_with_points` = fold + tile sort (the strengths
  travel inside the sorted records) -> spread -> pruned FFT passes (deconvolution
  fused). The same metric at every N, so that value(N) / (N value(1)) is the
  scaling efficiency.
N > 1 additionally (`config.config5_sharded`, after the timed region): BASELINE.json
  `configs[4]`, the batched workload north_star shards -- 2D type-1, 512 x 512,
  batch = 256 items of M = 1e6 points each, split over the N ranks in contiguous
  blocks (`tensorflow_nufft.sharding.shard_bounds`), every rank one `tfft.nufft`
  call per step over its block ("strong": 256 items in total), with its own
  roofline block, the result all_gather over RCCL beside it, the whole 256-item
  job on rank 0's GPU alone and `equal_work_efficiency` = sharded rate / (N x
  that). `--workload config5` makes that leg the line itself.

When `--gpus N` > 1 and the process was not started by torch.distributed.run
(no WORLD_SIZE in the environment), bench.py starts the N ranks itself as child
processes BEFORE anything touches the GPU, one device each, and exits with their
status.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline      the dominant kernel (spread) against the HBM roofline, timed with
                HIP events on the plan's stream inside the timed region, and
                `roofline.lds`: the same launches against the LDS-atomic pipe
  cpu_baseline  the CPU oracle (a port of the reference CPU path, NOT the
                upstream binary) on this host's cores, rank 0 at N = 1
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))

GRID = [1024, 1024]
M = 10_000_000
TOL = 1e-6
C5_GRID = [512, 512]
C5_ITEMS = 256
C5_M = 1_000_000
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec
LDS_ADD_F64_CYCLES = 8.6   # cycles per ds_add_f64 wave-instruction per CU, conflict free (profiles/r01_lds_atomic_ubench.txt)
LDS_ADD_U64_CYCLES = 7.0   # the same for ds_add_u64, the packed fixed-point accumulation of the 3-D float kernels
NUM_CUS = 256
CLOCK_GHZ_NOMINAL = 2.4


def shader_clock_ghz():
  """The clock the LDS roofline is priced at: measured on this device while every CU runs LDS atomics
  (nufft_hip_debug_shader_clock_mhz), not the nominal 2.4 GHz (r03's phase logs: ~2.05 GHz under this load)."""
  import ctypes
  import torch
  from tensorflow_nufft import _lib
  mhz = ctypes.c_double(0.0)
  rc = _lib.lib().nufft_hip_debug_shader_clock_mhz(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.byref(mhz))
  if rc != 0 or not (500.0 < mhz.value < 4000.0):
    return CLOCK_GHZ_NOMINAL, 'nominal (measurement failed)'
  return mhz.value * 1e-3, 'measured under an LDS-atomic load (nufft_hip_debug_shader_clock_mhz)'


def algorithmic_spread_bytes(m, nf, rank):
  # read each point's coordinates + complex strength once,
  # write each fine-grid cell once: M (4 d + 8) + 8 nf^d   [fp32]
  cells = 1
  for n in nf:
    cells *= n
  return m * (4 * rank + 8) + 8 * cells


def algorithmic_step_bytes(m, nf, nmodes, rank):
  # SURVEY.md section 8(d), "Whole type-1 transform": the spread bytes + the sort (read d 4 M coordinates; the records are
  # physically reordered: (4 d + 8) M written and read again) + the FFT (2 x 8 nf^d per pass, one pass per dimension) +
  # the deconvolution (8 N^d read + 8 N^d written).  cfg2: 193.6 + 400 + 134.2 + 16.8 = 744.6 MB.
  cells, modes = 1, 1
  for n in nf:
    cells *= n
  for n in nmodes:
    modes *= n
  return (algorithmic_spread_bytes(m, nf, rank) + 4 * rank * m + 2 * (4 * rank + 8) * m + 2 * 8 * cells * rank + 16 * modes)


# dominant kernel of the headline config (2-D, w = 8, float, >= 0.5 points per fine cell)
SPREAD_KERNEL = 'spread_2d_w8_group_kernel'


def pmc_traffic(kernel_name, m):
  """HBM-side bytes per launch of the spread kernel from the PMC passes committed
  under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs,
  corrected as MI355X_MICROARCH.md prescribes: 2*FETCH_SIZE + WRITE_SIZE).
  bench.py cannot run the profiler on itself, so the figure is the one measured
  offline for the same kernel and point count (`traffic_source` says so); None
  when it does not apply."""
  try:
    d = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_spread_traffic.json')))
    if d.get('points') == m and kernel_name in d.get('kernel', ''):
      return int(d['traffic_bytes_corrected']), d.get('source', 'profiles/pmc_spread_traffic.json')
  except (OSError, ValueError, KeyError):
    pass
  return None, None


def pmc_traffic_of(config_key):
  """The same for the dominant kernels of configs 3, 4, 5 (profiles/pmc_traffic_configs.json: one entry per
  config with the kernel, the corrected and raw byte counts and the profile file they come from)."""
  try:
    d = json.load(open(os.path.join(ROOT, 'profiles', 'pmc_traffic_configs.json'))).get(config_key)
    if d:
      return {'traffic': int(d['traffic_bytes_corrected']), 'traffic_raw': int(d.get('traffic_bytes_raw', 0)) or None,
              # (gathers counted at their 64-byte sectors, streams doubled: profiles/r04_fetch_calibration.txt)
              'traffic_bytes_calibrated': int(d.get('traffic_bytes_calibrated', 0)) or None,
              'traffic_kernel': d.get('kernel'), 'traffic_source': 'offline: ' + d.get('source', 'profiles/')}
  except (OSError, ValueError, KeyError, TypeError):
    pass
  return {'traffic': None}


def effective_cpus():
  """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU boxes of this pool
  show 256 hardware threads and a quota of 16 CPUs: more OpenMP threads than that are throttled, which is what made the
  CPU baseline 'peak at 32 of 256 threads' in r01-r03)."""
  try:
    n = len(os.sched_getaffinity(0))
  except AttributeError:
    n = os.cpu_count() or 1
  quota = None
  try:
    q, p = open('/sys/fs/cgroup/cpu.max').read().split()[:2]          # cgroup v2
    if q != 'max':
      quota = int(q) / int(p)
  except (OSError, ValueError):
    try:                                                               # cgroup v1
      q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
      p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
      if q > 0:
        quota = q / p
    except (OSError, ValueError):
      pass
  if quota:
    n = max(1, min(n, int(quota + 0.5)))
  return n, quota


def cpu_baseline(args):
  """Times the CPU oracle (port of the reference CPU path) on a bounded sample of the
  same workload: median of >= 5 runs after a warm-up, full transform and spreader
  alone, over a few OpenMP thread counts (the best median is reported)."""
  import numpy as np
  from oracle import oracle
  hw = os.cpu_count() or 1
  cores, quota = effective_cpus()
  m = args.cpu_points
  rng = np.random.default_rng(2)
  pts = rng.uniform(-np.pi, np.pi, (m, 2)).astype(np.float32)
  c = (rng.uniform(-.5, .5, m) + 1j * rng.uniform(-.5, .5, m)).astype(np.complex64)
  tol = float(np.float32(TOL))
  # reference CPU rule for this config: sigma = 1.25, w = 10,
  # float arithmetic, piecewise-polynomial kernel (kerevalmeth 1). Above 10 threads the
  # subgrids merge with atomics, below under a critical section (nufft_plan.cc:1109-1114).
  # thread counts around what the process may use (a little above it too: SMT siblings, quota granularity)
  cand = sorted({max(1, cores // 2), max(1, (3 * cores) // 4), cores, min(hw, (3 * cores) // 2), min(hw, 2 * cores)})
  budget = time.perf_counter() + 40.0
  best = None
  for nt in cand:
    # the C entry alone, on inputs already in its layout (r04: the numpy transpose of the points inside oracle.nufft,
    # ~50 ms on one thread at M = 1e7, used to sit in the timed region and capped the rate near 100 Mpts/s)
    full = oracle.time_nufft(c, pts, GRID, tol=TOL, kerevalmeth=1, nthreads=nt, repeats=5)
    spread = [oracle.time_spread(c, pts, GRID, tol=tol, sigma=0.0, kerevalmeth=1, nthreads=nt) for _ in range(5)]
    med_full, med_spread = float(np.median(full)), float(np.median(spread))
    if best is None or med_full < best[0]:
      best = (med_full, med_spread, nt, sorted(full))
    if time.perf_counter() > budget:
      break
  med_full, med_spread, nt, runs = best
  sigma, w, _, nf = oracle.query(2, GRID, tol, 'f32')
  # The oracle's FFT is its own radix code, the reference links FFTW: time both that and a
  # library FFT (scipy's pocketfft, same thread count) on the fine grid, and report the full
  # transform with the library FFT in its place alongside.
  lib_fft = None
  try:
    import scipy.fft
    a = (rng.uniform(-.5, .5, (nf[1], nf[0])) + 1j * rng.uniform(-.5, .5, (nf[1], nf[0]))).astype(np.complex64)
    t_own, t_lib = [], []
    for _ in range(5):
      b = a.copy(); t0 = time.perf_counter(); oracle.fft(b, -1, nthreads=nt); t_own.append(time.perf_counter() - t0)
      t0 = time.perf_counter(); scipy.fft.fft2(a, workers=nt); t_lib.append(time.perf_counter() - t0)
    own, libt = float(np.median(t_own)), float(np.median(t_lib))
    lib_fft = {'oracle_fft_ms': round(own * 1e3, 2), 'pocketfft_ms': round(libt * 1e3, 2),
               'value_with_library_fft_Mpts_s': round(m / max(med_full - own + libt, 1e-9) / 1e6, 3)}
  except Exception as e:   # scipy missing / oracle built without oracle_fft: the plain number stands
    lib_fft = {'error': str(e)[:100]}
  return {
      'fft_leg': lib_fft,
      'value': round(m / med_full / 1e6, 3), 'unit': 'Mpts/s', 'cores': nt, 'kind': 'port',
      'spread_only_Mpts_s': round(m / med_spread / 1e6, 3),
      'sample': f'{m} of the {M} points on the same 1024x1024 grid, reference CPU rule sigma={sigma} w={w} '
                f'fine grid {nf[0]}x{nf[1]}, fp32; value = full type-1 transform (sort + spread + FFT + '
                f'deconvolve: the C entry alone on coordinate arrays), spread_only = the spreader on pre-sorted points; median of 5 runs after a '
                f'warm-up at the best of the thread counts {cand}: {nt} OpenMP threads; this process may use {cores} CPUs '
                f'(cgroup quota {quota if quota else "none"}, {hw} hardware threads on the host); '
                f'full-transform runs {[round(r, 3) for r in runs]} s',
      'usable_cpus': cores, 'hardware_threads': hw,
  }


def lds_roofline(pts, info, spread_ms):
  """The spread launches against the LDS-atomic pipe: ds_add_f64 wave-instructions per
  launch / kernel time, against CUs x clock / 8.6 cycles. The cell-grouped kernel issues
  one (re, im) pair per run of points sharing a stencil start cell; the runs are counted
  here from the points (distinct start cells + the forced run ends every 32 staged
  points), not assumed."""
  import numpy as np
  import torch
  nf = [int(info.fine_dims[1]), int(info.fine_dims[0])]
  w = int(info.kernel_width)
  p = pts.to(torch.float64)
  keys = None
  for d in range(2):
    xs = (p[:, d] + np.pi) * (nf[d] / (2 * np.pi))
    i0 = torch.ceil(xs - w / 2).to(torch.int64) % nf[d]
    keys = i0 if keys is None else keys * nf[d] + i0
  groups = int(torch.unique(keys).numel())
  m = pts.shape[0]
  groups += (m - groups) // 32      # a run also ends where the 32-point staging block ends
  instr = 2.0 * groups
  ghz, how = shader_clock_ghz()
  peak = NUM_CUS * ghz * 1e9 / LDS_ADD_F64_CYCLES
  achieved = instr / (spread_ms * 1e-3)
  return {
      'bound': 'lds-atomic', 'achieved': round(achieved / 1e9, 2), 'peak': round(peak / 1e9, 2),
      'unit': 'G wave-instr/s (ds_add_f64)', 'frac': round(achieved / peak, 4),
      'atomics_per_point': round(instr / m, 3), 'clock_ghz': round(ghz, 3), 'clock': how,
      'note': 'peak = 256 CUs x clock_ghz / 8.6 cycles per conflict-free ds_add_f64 wave-instruction '
              '(tools/ubench/lds_atomic_bench.hip); achieved counts only the atomics, the same pipe also '
              'serves 0.75 ds_read_b128 + 0.375 ds_write_b32 wave-instructions per point',
  }


def _event_timed(fn, steps):
  """ms per call of fn over `steps` calls, HIP events on the current stream (the stream the
  plans and tfft.nufft launch on)."""
  import torch
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  torch.cuda.synchronize()
  e0.record()
  for _ in range(steps):
    fn()
  e1.record()
  e1.synchronize()
  return e0.elapsed_time(e1) / steps


def other_configs(args, dev):
  """BASELINE configs 3, 4 and 5 on this one GPU, after the headline region: ms per step (HIP
  events), whole-step rate, the dominant kernel's average duration (HIP events of the plan's
  own stage timing around that kernel) and its HBM fraction on the SURVEY 8(d) algorithmic
  bytes; the 3-D type-1 entries also carry `lds`: the spread kernel's ds_add_u64 rate against
  the LDS-atomic peak at the measured clock (the bound that tracks those kernels).
  `value` stays config 2; these are the driver-timed figures of the other configs."""
  import numpy as np
  import torch
  import tensorflow_nufft as tfft
  out = {}

  def rnd_c(shape, g):
    return torch.complex(torch.rand(shape, generator=g, device=dev) - .5, torch.rand(shape, generator=g, device=dev) - .5)

  def radial_points(m, rank, g):
    """MRI-like trajectories in acquisition order (SURVEY 8d's secondary stress input; docs/examples/mri_app.ipynb
    upstream): spokes through the centre, 1000 (2-D, golden-angle) / 500 (3-D kooshball, random directions) samples each."""
    ns = 1000 if rank == 2 else 500
    nsp = m // ns
    s = torch.linspace(-np.pi, np.pi, ns + 1, device=dev)[:ns]
    if rank == 2:
      a = torch.arange(nsp, device=dev) * (np.pi * 0.6180339887)
      d = torch.stack([torch.cos(a), torch.sin(a)], dim=1)
    else:
      u = torch.rand(nsp, generator=g, device=dev) * 2 - 1
      ph = torch.rand(nsp, generator=g, device=dev) * 2 * np.pi
      d = torch.stack([torch.sqrt(1 - u * u) * torch.cos(ph), torch.sqrt(1 - u * u) * torch.sin(ph), u], dim=1)
    return (d[:, None, :] * s[None, :, None]).reshape(-1, rank).contiguous()

  def plan_case(name, ttype, grid, m, tol, seed, steps, stage, dist='uniform', cdtype=torch.complex64, also=()):
    g = torch.Generator(device=dev).manual_seed(seed)
    rank = len(grid)
    rdtype = torch.float32 if cdtype == torch.complex64 else torch.float64
    if dist == 'radial':
      pts = radial_points(m, rank, g)
      m = int(pts.shape[0])
    else:
      pts = (torch.rand((m, rank), generator=g, device=dev, dtype=rdtype) * 2 - 1) * np.pi
      if dist == 'blob':   # uniform background + 20 % of the points in a Gaussian blob of sigma = 3 fine cells (tools/ab_fallback_group.py)
        k = m // 5
        pts[:k] = 0.4 + torch.randn((k, rank), generator=g, device=dev, dtype=rdtype) * (3 * np.pi / grid[0])
        pts = pts[torch.randperm(m, device=dev, generator=g)].contiguous()
    src = rnd_c([m] if ttype == 'type_1' else grid, g).to(cdtype)
    plan = tfft.Plan(ttype, grid, 'forward', tol=tol, dtype=cdtype, device=dev)
    info = plan.info()
    res = torch.empty(grid if ttype == 'type_1' else [m], dtype=cdtype, device=dev)
    step = lambda: plan.execute_with_points(pts, src, out=res)
    for _ in range(2):
      step()
    ms = _event_timed(step, steps)
    plan.set_timing(2)
    plan.get_timing()
    for _ in range(steps):
      step()
    tm = plan.get_timing()
    plan.set_timing(False)
    k_ms = tm[stage][0] / max(tm[stage][1], 1)
    nf = [int(info.fine_dims[d]) for d in range(rank)]
    algo = algorithmic_spread_bytes(m, nf, rank)   # (type 2: the fine grid is read, the results written: same count)
    if cdtype == torch.complex128:
      algo *= 2                                    # (the fp64 form of SURVEY 8(d): M (8 d + 16) + 16 nf^d)
    out[name] = {'ms_per_step': round(ms, 4), 'Gpts_s': round(m / ms / 1e6, 2), 'dominant_kernel': stage,
                 'dominant_kernel_ms': round(k_ms, 4), 'algorithmic_bytes': algo,
                 'hbm_frac': round(algo / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                 'kernel_width': int(info.kernel_width), 'fine_grid': nf[::-1], 'steps': steps, 'points': dist,
                 'dtype': 'f64' if cdtype == torch.complex128 else 'f32'}
    for st in also:   # further stages of the same pass (timing level 2 covers the dominant kernel only: a third pass)
      if not any(tm.get(k, (0, 0))[1] for k in {'fft': ['fft'], 'sort': ['sort_scatter']}.get(st, [st])):
        plan.set_timing(1); plan.get_timing()
        for _ in range(steps):
          step()
        tm = plan.get_timing()
        plan.set_timing(False)
      groups = {'fft': ['fft', 'deconvolve', 'zero'], 'sort': ['sort_count', 'sort_scan', 'sort_scatter']}
      out[name][st + '_ms'] = round(sum(tm[k][0] / max(tm[k][1], 1) for k in groups.get(st, [st]) if k in tm), 4)
      if st == 'fft' and ttype == 'type_1' and cdtype == torch.complex64:
        # the pruned type-1 passes against HBM: pass d reads what pass d - 1 kept and writes half of it (sigma = 2), the first
        # one also stores zeros over the fine grid it read: 8 B x cells x (1 + 1 + 1/2 + 1/2 + 1/4 + 1/4 + 1/8 ...) 
        cells = 1
        for n in nf:
          cells *= n
        fb, kept = 8 * cells, float(cells)      # (the zeroing store)
        for d in range(rank):
          fb += 8 * kept * 1.5
          kept *= 0.5
        out[name]['fft_algorithmic_bytes'] = int(fb)
        out[name]['fft_hbm_frac'] = round(fb / (out[name]['fft_ms'] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    if ttype == 'type_1' and rank == 3 and int(info.kernel_width) <= 8 and cdtype == torch.complex64:
      # the 3-D float fixed-point spreaders are bound by the LDS-atomic data path, not by HBM: ds_add_u64
      # wave-instructions per point of the kernel family (spread_dense3_kernel: one W x 3 x 3 lane block per
      # instruction, 1 at w <= 4 and 4 at w = 5, 6; spread_patch3_kernel: one 8 x 8 (x, y) patch per z plane, w of them)
      w_ = int(info.kernel_width)
      per_pt = 1 if w_ <= 4 else (4 if w_ <= 6 else w_)
      ghz, how = shader_clock_ghz()
      peak = NUM_CUS * ghz * 1e9 / LDS_ADD_U64_CYCLES
      achieved = m * per_pt / (k_ms * 1e-3)
      out[name]['lds'] = {'bound': 'lds-atomic', 'unit': 'G wave-instr/s (ds_add_u64)', 'achieved': round(achieved / 1e9, 2),
                          'peak': round(peak / 1e9, 2), 'frac': round(achieved / peak, 4), 'atomics_per_point': per_pt,
                          'clock_ghz': round(ghz, 3), 'clock': how}
    plan.close()
    del pts, src, res
    torch.cuda.empty_cache()

  def guarded(name, fn):
    # (a memory or time failure in one of these legs must not lose the headline line)
    try:
      fn()
    except Exception as e:   # pylint: disable=broad-except
      out[name] = {'error': f'{type(e).__name__}: {str(e)[:200]}'}
      torch.cuda.empty_cache()

  guarded('config3_2d_type2_1024_M1e7',
          lambda: plan_case('config3_2d_type2_1024_M1e7', 'type_2', GRID, M, TOL, 3, max(5, args.steps // 2), 'interp'))
  guarded('config4_3d_type1_256_M1e8_tol1e-4',
          lambda: plan_case('config4_3d_type1_256_M1e8_tol1e-4', 'type_1', [256, 256, 256], 100_000_000, 1e-4, 4, 5, 'spread'))
  # the reference harness's own 3-D case at the API's default tolerance, scaled to config 4's grid (w = 8)
  guarded('3d_type1_256_M3e7_tol1e-6',
          lambda: plan_case('3d_type1_256_M3e7_tol1e-6', 'type_1', [256, 256, 256], 30_000_000, 1e-6, 6, 5, 'spread'))
  # radial point sets (acquisition order) at the sizes above: the uniform entries are the ones to read them against
  guarded('radial_2d_type1_1024_M1e7',
          lambda: plan_case('radial_2d_type1_1024_M1e7', 'type_1', GRID, M, TOL, 7, max(5, args.steps // 2), 'spread', 'radial'))
  guarded('radial_2d_type2_1024_M1e7',
          lambda: plan_case('radial_2d_type2_1024_M1e7', 'type_2', GRID, M, TOL, 8, max(5, args.steps // 2), 'interp', 'radial'))
  guarded('kooshball_3d_type1_256_M3e7_tol1e-6',
          lambda: plan_case('kooshball_3d_type1_256_M3e7_tol1e-6', 'type_1', [256, 256, 256], 30_000_000, 1e-6, 9, 5, 'spread', 'radial'))
  # r06 (r05 verdict): what r05 / r06 built, under the driver's clock -- stacks of tiles (w = 8 and w = 6), the 3-D type-2
  # tile loader, the cell-grouped fp64 fallback on a clustered set, a fine grid that is not a power of two (mixed-radix
  # pruned passes; `fft_ms` = FFT + deconvolve + zero stages), a complex128 transform (fp64 planes over stacks)
  guarded('3d_type1_256_M1e7_tol1e-6_stacks',
          lambda: plan_case('3d_type1_256_M1e7_tol1e-6_stacks', 'type_1', [256, 256, 256], 10_000_000, 1e-6, 10, 5, 'spread', also=('fft',)))
  guarded('3d_type1_256_M1e7_tol1e-4_dense_stacks',
          lambda: plan_case('3d_type1_256_M1e7_tol1e-4_dense_stacks', 'type_1', [256, 256, 256], 10_000_000, 1e-4, 11, 5, 'spread'))
  guarded('3d_type2_256_M1e7_tol1e-6',
          lambda: plan_case('3d_type2_256_M1e7_tol1e-6', 'type_2', [256, 256, 256], 10_000_000, 1e-6, 12, 5, 'interp'))
  guarded('uniform_plus_blob_3d_type1_256_M3e7_tol1e-6',
          lambda: plan_case('uniform_plus_blob_3d_type1_256_M3e7_tol1e-6', 'type_1', [256, 256, 256], 30_000_000, 1e-6, 13, 3, 'spread', 'blob'))
  guarded('nonpow2_3d_type1_240_M1e7_tol1e-6',
          lambda: plan_case('nonpow2_3d_type1_240_M1e7_tol1e-6', 'type_1', [240, 240, 240], 10_000_000, 1e-6, 14, 5, 'spread', also=('fft',)))
  guarded('nonpow2_2d_type1_960_M1e7_tol1e-6',
          lambda: plan_case('nonpow2_2d_type1_960_M1e7_tol1e-6', 'type_1', [960, 960], M, TOL, 15, 5, 'spread', also=('fft',)))
  guarded('c128_3d_type1_256_M1e7_tol1e-6',
          lambda: plan_case('c128_3d_type1_256_M1e7_tol1e-6', 'type_1', [256, 256, 256], 10_000_000, 1e-6, 16, 3, 'spread', cdtype=torch.complex128))
  guarded('c128_3d_type2_256_M1e7_tol1e-6',
          lambda: plan_case('c128_3d_type2_256_M1e7_tol1e-6', 'type_2', [256, 256, 256], 10_000_000, 1e-6, 18, 3, 'interp', cdtype=torch.complex128))
  # a grid past 640^3 fine cells (two-level sort over 1728 super-tiles; `sort_ms` = count + scan + scatter)
  guarded('3d_type1_384_M1e8_tol1e-4',
          lambda: plan_case('3d_type1_384_M1e8_tol1e-4', 'type_1', [384, 384, 384], 100_000_000, 1e-4, 19, 3, 'spread', also=('sort',)))
  for k in list(out):
    if 'error' not in out[k]:
      out[k].update(pmc_traffic_of(k))

  # the reference harness's own cases (NUFFTOpsBenchmark.cases, nufft_ops_test.py:732-741: its first 2-D case and its
  # two 3-D ones), through tfft.nufft at the API's default tolerance like the harness: ms per call, HIP events
  def op_case(name, source_shape, points_shape, ttype, grid):
    g = torch.Generator(device=dev).manual_seed(17)
    src = rnd_c(source_shape, g)
    pts = (torch.rand(points_shape, generator=g, device=dev) - .5) * (2.0 * np.pi)
    call = lambda: tfft.nufft(src, pts, grid_shape=grid, transform_type=ttype)
    for _ in range(3):
      call()
    ms = _event_timed(call, 20)
    npts = 1
    for v in points_shape[:-1]:
      npts *= v
    out[name] = {'ms_per_call': round(ms, 4), 'Gpts_s': round(npts / ms / 1e6, 3), 'through': 'tfft.nufft (op-level entry, cached plan)',
                 'source_shape': source_shape, 'points_shape': points_shape}
  guarded('refharness_2d_type2_256_M2e5', lambda: op_case('refharness_2d_type2_256_M2e5', [256, 256], [200000, 2], 'type_2', None))
  guarded('refharness_3d_type2_128_M8e5', lambda: op_case('refharness_3d_type2_128_M8e5', [128, 128, 128], [800000, 3], 'type_2', None))
  guarded('refharness_3d_type1_128_M8e5', lambda: op_case('refharness_3d_type1_128_M8e5', [800000], [800000, 3], 'type_1', [128, 128, 128]))
  try:
    out.update(config5_one_gpu(args, dev))
  except Exception as e:   # pylint: disable=broad-except
    out['config5_batched_2d_type1_512_32_items'] = {'error': f'{type(e).__name__}: {str(e)[:200]}'}
  return out


def config5_spread_kernel(dev, pts, c, grp=16):
  """Dominant kernel of config 5 as the op runs it (one plan, num_point_sets = 16): per-item spread time from the
  plan's own HIP events, algorithmic bytes per item, HBM fraction."""
  import torch
  import tensorflow_nufft as tfft
  grp = min(grp, pts.shape[0])
  plan = tfft.Plan('type_1', C5_GRID, 'forward', tol=TOL, dtype=torch.complex64, device=dev, num_point_sets=grp)
  info = plan.info()
  pg, cg = pts[:grp].contiguous(), c[:grp].contiguous()
  if grp == 1:
    pg, cg = pg[0], cg[0]
  for _ in range(2):
    plan.execute_with_points(pg, cg)
  plan.set_timing(2)
  plan.get_timing()
  for _ in range(5):
    plan.execute_with_points(pg, cg)
  tm = plan.get_timing()
  plan.close()
  k_ms = tm['spread'][0] / max(tm['spread'][1], 1) / grp     # per item
  nf = [int(info.fine_dims[d]) for d in range(2)]
  algo = algorithmic_spread_bytes(C5_M, nf, 2)
  return {'dominant_kernel': 'spread', 'dominant_kernel_ms_per_item': round(k_ms, 4), 'algorithmic_bytes_per_item': algo,
          'hbm_frac': round(algo / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), 'kernel_width': int(info.kernel_width),
          'fine_grid': nf[::-1], **pmc_traffic_of('config5_item')}


def config5_one_gpu(args, dev):
  import numpy as np
  import torch
  import tensorflow_nufft as tfft
  out = {}

  def rnd_c(shape, g):
    return torch.complex(torch.rand(shape, generator=g, device=dev) - .5, torch.rand(shape, generator=g, device=dev) - .5)

  # config 5: one GPU's share at N = 8 (32 items), and the whole 256-item job on this GPU (the
  # N = 1 anchor of the scaling curve), both through tfft.nufft with per-item points
  g = torch.Generator(device=dev).manual_seed(5)
  share = 32
  pts = (torch.rand((share, C5_M, 2), generator=g, device=dev) * 2 - 1) * np.pi
  c = rnd_c((share, C5_M), g)
  call = lambda: tfft.nufft(c, pts, grid_shape=C5_GRID, transform_type='type_1', tol=TOL)
  for _ in range(2):
    call()
  ms32 = _event_timed(call, max(5, args.steps // 5))
  def whole():
    for _ in range(C5_ITEMS // share):
      call()
  ms256 = _event_timed(whole, 3)
  c5 = config5_spread_kernel(dev, pts, c)
  out['config5_batched_2d_type1_512_32_items'] = dict(c5, ms_per_step=round(ms32, 4), Gpts_s=round(share * C5_M / ms32 / 1e6, 2))
  out['config5_whole_job_256_items_one_gpu'] = dict(c5, ms_per_step=round(ms256, 4), Gpts_s=round(C5_ITEMS * C5_M / ms256 / 1e6, 2),
                                                     note='the N = 1 anchor for the sharded runs (bench.py --gpus N reports this workload)')
  return out


def spawn_ranks(args, argv):
  """--gpus N > 1 without a launcher: one child process per rank, started before any GPU
  call in this process (a process that touched the GPU must not exec or fork workers)."""
  # Rendezvous through a FILE store in a private temporary directory (nobody can take it between its creation
  # and the children's use, unlike a "free" TCP port picked here and bound later); an explicit MASTER_PORT in the
  # environment is honoured as before.
  import shutil
  import tempfile
  tmpdir = None
  extra = {}
  if 'MASTER_PORT' in os.environ:
    extra = {'MASTER_ADDR': os.environ.get('MASTER_ADDR', '127.0.0.1'), 'MASTER_PORT': os.environ['MASTER_PORT']}
  else:
    tmpdir = tempfile.mkdtemp(prefix='nufft_bench_')
    extra = {'NUFFT_BENCH_INIT_FILE': os.path.join(tmpdir, 'store')}
  procs = []
  try:
    for r in range(args.gpus):
      env = dict(os.environ)
      env.update({'RANK': str(r), 'LOCAL_RANK': str(r), 'WORLD_SIZE': str(args.gpus),
                  'LOCAL_WORLD_SIZE': str(args.gpus),
                  'HSA_ENABLE_IPC_MODE_LEGACY': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'), **extra})
      procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    for p in procs:
      p.wait()
      rc = rc or p.returncode
    return rc
  finally:
    if tmpdir:
      shutil.rmtree(tmpdir, ignore_errors=True)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=50)
  ap.add_argument('--warmup', type=int, default=5)
  ap.add_argument('--points', type=int, default=M)
  ap.add_argument('--cpu-points', type=int, default=M)
  ap.add_argument('--items', type=int, default=C5_ITEMS, help='batch size of the sharded workload (N > 1)')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-extras', action='store_true',
                  help='skip the informational legs (used under rocprofv3 so that its per-kernel '
                       'average covers the timed launches only)')
  ap.add_argument('--no-other-configs', action='store_true', help='skip config.other_configs (configs 3, 4, 5 on this GPU)')
  ap.add_argument('--dist-backend', default='nccl', help='nccl (= RCCL, default) | gloo (testing)')
  ap.add_argument('--device', type=int, default=None, help='force a device index (testing)')
  ap.add_argument('--force-dist', action='store_true', help='initialise torch.distributed even at world size 1 (testing)')
  ap.add_argument('--workload', default='auto', help='auto | config2 | config5 (testing: config5 on one GPU)')
  args = ap.parse_args()

  if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
    sys.exit(spawn_ranks(args, sys.argv[1:]))

  # The JSON line must be the only (and last) thing on stdout. Libraries below write there through C stdio -- RCCL's version
  # banner is flushed at process exit, i.e. BEHIND anything Python printed -- so file descriptor 1 is pointed at stderr for the
  # life of the process and the line goes out through a private duplicate of the real stdout.
  sys.stdout.flush()
  json_fd = os.dup(1)
  os.dup2(2, 1)
  import numpy as np   # noqa: F401
  import torch
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
    os.environ.setdefault('RANK', '0')
    os.environ.setdefault('WORLD_SIZE', '1')
    kw = {}
    init_file = os.environ.get('NUFFT_BENCH_INIT_FILE')
    tmp_store = None
    if 'MASTER_PORT' not in os.environ and not init_file:   # (--force-dist at world size 1: a private file store)
      import tempfile
      tmp_store = tempfile.mkdtemp(prefix='nufft_bench_')
      init_file = os.path.join(tmp_store, 'store')
    if init_file and 'MASTER_PORT' not in os.environ:
      kw = {'init_method': 'file://' + init_file, 'rank': rank, 'world_size': world}
    else:
      os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    if args.dist_backend == 'nccl':
      dist.init_process_group('nccl', device_id=dev, **kw)   # "nccl" is RCCL on ROCm
    else:
      dist.init_process_group(args.dist_backend, **kw)
  workload = args.workload
  if workload == 'auto':
    workload = 'config2'
  if workload == 'config5':
    result = run_config5(args, dev, dist, world, rank)
  else:
    result = run_config2(args, dev, dist, world, rank)
    if world > 1 and not args.no_extras:
      # the batched workload north_star shards, on the same ranks. A rank-local failure inside it (allocation, plan
      # creation) must not leave the other ranks waiting in a collective with the headline line unprinted: the line
      # goes to stderr first as a breadcrumb, every rank runs its share ONCE without any collective, and the ranks
      # agree that all of them got through before the timed, collective-bearing leg starts.
      if rank == 0:
        print('headline (before the sharded config-5 leg): ' + json.dumps(result), file=sys.stderr, flush=True)
      c5, ok = None, 1
      try:
        prep = prepare_config5(args, dev, world, rank)
      except Exception as e:   # pylint: disable=broad-except
        prep, ok = None, 0
        c5 = {'error': f'rank {rank}: {type(e).__name__}: {str(e)[:200]}'}
      flag = torch.tensor([ok], dtype=torch.int32, device=dev if args.dist_backend == 'nccl' else 'cpu')
      dist.all_reduce(flag, op=dist.ReduceOp.MIN)
      if int(flag.item()) == 1:
        try:
          c5 = run_config5(args, dev, dist, world, rank, prep)
        except Exception as e:   # pylint: disable=broad-except
          c5 = {'error': f'{type(e).__name__}: {str(e)[:200]}'}
      elif c5 is None:
        c5 = {'error': 'another rank failed to set the sharded workload up; leg skipped on every rank'}
      if rank == 0:
        result['config']['config5_sharded'] = c5
        # (what the driver's record should show without digging)
        cfg5 = c5.get('config', {}) if isinstance(c5, dict) else {}
        result['config']['equal_work_efficiency'] = cfg5.get('equal_work_efficiency')
        result['config']['rccl_world_size'] = cfg5.get('rccl_world_size')
  if rank == 0:
    os.write(json_fd, (json.dumps(result) + '\n').encode())
  os.close(json_fd)
  if dist is not None:
    dist.destroy_process_group()


def timed(fn, steps, dist, dev, backend):
  """EXACTLY `steps` calls of fn bracketed by barrier + synchronize; max over ranks."""
  import torch
  if dist is not None:
    dist.barrier()
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for _ in range(steps):
    fn()
  torch.cuda.synchronize()
  if dist is not None:
    dist.barrier()
  elapsed = time.perf_counter() - t0
  if dist is not None:
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == 'nccl' else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
  return elapsed


def run_config2(args, dev, dist, world, rank):
  import numpy as np
  import torch
  import tensorflow_nufft as tfft
  m = args.points
  # synthetic inputs: points ~ U(-pi, pi)^2,
  # strengths ~ U(-.5,.5) + i U(-.5,.5); seed 2 (+rank so replicas differ)
  g = torch.Generator(device=dev).manual_seed(2 + rank)
  pts = (torch.rand((m, 2), generator=g, device=dev) * 2 - 1) * np.pi
  c = torch.complex(torch.rand(m, generator=g, device=dev) - .5,
                    torch.rand(m, generator=g, device=dev) - .5)
  plan = tfft.Plan('type_1', GRID, 'forward', tol=TOL, dtype=torch.complex64, device=dev)
  info = plan.info()
  out = torch.empty(GRID, dtype=torch.complex64, device=dev)

  def step():
    plan.execute_with_points(pts, c, out=out)

  for _ in range(args.warmup):
    step()
  # full per-stage breakdown from an untimed pass (14 events per step cost ~7 %). The device's clocks settle only
  # after ~10 ms of work -- a timed region that starts after 6 steps measured 2-3 % below the sustained rate
  # (EXPERIMENTS.md section 10.19) -- so the W warm-up steps and this pass together are at least 20 steps;
  # `warmup_effective` in the line says how many untimed steps ran before the timed K.
  plan.set_timing(1)
  plan.get_timing()
  stage_steps = max(3, 20 - args.warmup)   # (W >= 20: the W steps are the warm-up and this pass is 3 steps long)
  for _ in range(stage_steps):
    step()
  stage_all = plan.get_timing()
  # timed region: HIP events around the dominant (spread) kernel only, recorded
  # on the plan's stream, i.e. the stream the kernel is launched on
  plan.set_timing(2)
  plan.get_timing()
  elapsed = timed(step, args.steps, dist, dev, args.dist_backend)
  stages = plan.get_timing()
  plan.set_timing(False)

  lds_stats = lds_roofline(pts, info, stages['spread'][0] / max(stages['spread'][1], 1)) if rank == 0 else None
  extras = {}
  if not args.no_extras:
    # the two-call form (set_points, then execute with the strengths gathered through the
    # sort permutation) and the exec-only rate of a plan whose points stay set
    def step2():
      plan.set_points(pts)
      plan.execute(c, out=out)
    for _ in range(3):
      step2()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
      step2()
    torch.cuda.synchronize()
    extras['two_call_Mpts_s'] = round(m / ((time.perf_counter() - t1) / args.steps) / 1e6, 2)
    t1 = time.perf_counter()
    for _ in range(args.steps):
      plan.execute(c, out=out)
    torch.cuda.synchronize()
    extras['exec_only_Mpts_s'] = round(m / ((time.perf_counter() - t1) / args.steps) / 1e6, 2)
    # the drop-in surface itself (op-level entry: validation, plan cache, fused call)
    for _ in range(3):
      tfft.nufft(c, pts, grid_shape=GRID, transform_type='type_1', tol=TOL)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(args.steps):
      tfft.nufft(c, pts, grid_shape=GRID, transform_type='type_1', tol=TOL)
    torch.cuda.synchronize()
    extras['tfft_nufft_Mpts_s'] = round(m / ((time.perf_counter() - t1) / args.steps) / 1e6, 2)

  if rank != 0:
    return None
  if world == 1 and not args.no_extras and not args.no_other_configs:
    plan.close()
    extras['other_configs'] = other_configs(args, dev)
  ms_per_step = elapsed / args.steps * 1e3
  value = world * m / (elapsed / args.steps) / 1e6
  nf = [int(info.fine_dims[1]), int(info.fine_dims[0])]
  spread_ms = stages['spread'][0] / max(stages['spread'][1], 1)
  algo = algorithmic_spread_bytes(m, nf, 2)
  achieved = algo / (spread_ms * 1e-3) / 1e9
  traffic, traffic_src = pmc_traffic(SPREAD_KERNEL, m)
  result = {
      'metric': 'non-uniform pts/s, 2D type-1 1024^2 tol=1e-6 (set_points + execute)',
      'value': round(value, 2), 'unit': 'Mpts/s', 'n_gpus': world, 'steps': args.steps,
      'warmup': args.warmup, 'warmup_effective': args.warmup + stage_steps,
      'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True,
      'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
      'config': {
          'workload': 'BASELINE configs[1]: 2D type-1, 1024x1024 modes, M=1e7 uniform random points, '
                      'tol=1e-6, complex64; one transform per rank and step through nufft_hip_execute_with_points '
                      '(the call tfft.nufft makes); ranks hold independent transforms (no collective in the data path)',
          'points_per_gpu': m, 'grid': GRID, 'fine_grid': nf, 'kernel_width': int(info.kernel_width),
          'upsampling_factor': info.upsampling_factor, 'spread_method': int(info.spread_method),
          'stage_us': {k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in stage_all.items() if v[1]},
          'untimed_steps': args.warmup + stage_steps,   # the W warm-up steps + the per-stage timing pass, before the timed K
          **extras,
      },
      'roofline': {
          'bound': 'hbm', 'kernel': SPREAD_KERNEL, 'achieved': round(achieved, 1),
          'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 4),
          'traffic': traffic, 'traffic_source': 'offline: ' + traffic_src if traffic else None,
          'algorithmic_bytes': algo, 'kernel_ms': round(spread_ms, 4),
          'lds': lds_stats,
          'note': 'algorithmic bytes = M (4 d + 8) + 8 nf^d (SURVEY.md 8d). The kernel is bound by the LDS '
                  'pipe and its per-tile phases, not by HBM (EXPERIMENTS.md section 4): roofline.lds is the bound '
                  'that tracks it.',
      },
  }
  step_bytes = algorithmic_step_bytes(m, nf, GRID, 2)
  step_gbs = step_bytes / (ms_per_step * 1e-3) / 1e9
  result['step_roofline'] = {
      'bound': 'hbm', 'achieved': round(step_gbs, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(step_gbs / HBM_PEAK_GBS, 4),
      'algorithmic_bytes': step_bytes, 'ms': round(ms_per_step, 4),
      'note': 'the whole step (sort + spread + FFT passes with the deconvolution fused) on SURVEY.md 8(d)\'s whole-type-1-transform '
              'bytes: spread M (4 d + 8) + 8 nf^d, sort 4 d M read + (4 d + 8) M written and read again, FFT 2 x 8 nf^d per dimension, '
              'deconvolution 16 N^d. The sort is ~40 % of the step and its scatter is bound by write transactions (one 16-byte record per '
              'fabric write), the spread by the LDS pipe: neither by HBM bytes.',
  }
  if world == 1 and not args.no_cpu_baseline:
    result['cpu_baseline'] = cpu_baseline(args)
  return result


def prepare_config5(args, dev, world, rank):
  """This rank's share of the sharded workload, run once: inputs, plans, workspaces. No collective in here."""
  import numpy as np
  import torch
  import tensorflow_nufft as tfft
  from tensorflow_nufft import sharding
  items, m = args.items, C5_M
  lo, hi = sharding.shard_bounds(items, world, rank)
  nloc = hi - lo
  # per-item points; seed 5 (+ rank: every rank draws its own block)
  g = torch.Generator(device=dev).manual_seed(5 + rank)
  pts = (torch.rand((nloc, m, 2), generator=g, device=dev) * 2 - 1) * np.pi
  c = torch.complex(torch.rand((nloc, m), generator=g, device=dev) - .5,
                    torch.rand((nloc, m), generator=g, device=dev) - .5)
  tfft.nufft(c, pts, grid_shape=C5_GRID, transform_type='type_1', tol=TOL)
  torch.cuda.synchronize()
  return pts, c


def run_config5(args, dev, dist, world, rank, prep=None):
  import numpy as np
  import torch
  import tensorflow_nufft as tfft
  from tensorflow_nufft import sharding
  items, m = args.items, C5_M
  lo, hi = sharding.shard_bounds(items, world, rank)
  nloc = hi - lo
  pts, c = prep if prep is not None else prepare_config5(args, dev, world, rank)

  def transform(s, p):
    return tfft.nufft(s, p, grid_shape=C5_GRID, transform_type='type_1', tol=TOL)

  def step():
    transform(c, pts)

  for _ in range(args.warmup):
    step()
  elapsed = timed(step, args.steps, dist, dev, args.dist_backend)

  extras = {}
  if not args.no_extras:
    # the same with the results all-gathered onto every rank (RCCL over xGMI at N > 1)
    if dist is not None and args.dist_backend == 'nccl':
      def step_gather():
        sharding.gather_blocks(transform(c, pts), items)
      for _ in range(2):
        step_gather()
      tg = timed(step_gather, args.steps, dist, dev, args.dist_backend)
      extras['with_all_gather_Mpts_s'] = round(items * m / (tg / args.steps) / 1e6, 2)
    # points shared by the whole batch: one sort, nloc transforms
    shared = pts[0]
    for _ in range(2):
      transform(c, shared)
    ts = timed(lambda: transform(c, shared), args.steps, dist, dev, args.dist_backend)
    extras['shared_points_Mpts_s'] = round(items * m / (ts / args.steps) / 1e6, 2)
    if world > 1 and rank == 0:
      # the WHOLE 256-item job on this one GPU, for the scaling comparison on equal work
      torch.cuda.synchronize()
      g1 = torch.Generator(device=dev).manual_seed(105)
      chunk = min(32, items)
      pc = (torch.rand((chunk, m, 2), generator=g1, device=dev) * 2 - 1) * np.pi
      cc = torch.complex(torch.rand((chunk, m), generator=g1, device=dev) - .5,
                         torch.rand((chunk, m), generator=g1, device=dev) - .5)
      def whole():
        for _ in range(items // chunk):
          transform(cc, pc)
      whole()
      torch.cuda.synchronize()
      t1 = time.perf_counter()
      for _ in range(args.steps):
        whole()
      torch.cuda.synchronize()
      extras['one_gpu_whole_job_Mpts_s'] = round((items // chunk) * chunk * m / ((time.perf_counter() - t1) / args.steps) / 1e6, 2)
    if dist is not None:
      dist.barrier()

  # ranks that actually took part, counted by the collective library itself
  measured_world = 1
  if dist is not None:
    ones = torch.ones(1, dtype=torch.int32, device=dev if args.dist_backend == 'nccl' else 'cpu')
    dist.all_reduce(ones)
    measured_world = int(ones.item())
  if rank != 0:
    return None
  ms_per_step = elapsed / args.steps * 1e3
  value = items * m / (elapsed / args.steps) / 1e6
  if 'one_gpu_whole_job_Mpts_s' in extras and extras['one_gpu_whole_job_Mpts_s'] > 0:
    # sharded rate over N x the same 256-item job on ONE of these GPUs: the scaling efficiency on equal work
    extras['equal_work_efficiency'] = round(value / (world * extras['one_gpu_whole_job_Mpts_s']), 4)
  roofline = None
  if not args.no_extras:
    try:
      k = config5_spread_kernel(dev, pts, c)
      ach = k['algorithmic_bytes_per_item'] / (k['dominant_kernel_ms_per_item'] * 1e-3) / 1e9
      roofline = {'bound': 'hbm', 'kernel': SPREAD_KERNEL + ' (16 point sets per launch; per item)', 'achieved': round(ach, 1),
                  'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(ach / HBM_PEAK_GBS, 4), 'traffic': k.get('traffic'),
                  'traffic_source': k.get('traffic_source'), 'algorithmic_bytes': k['algorithmic_bytes_per_item'],
                  'kernel_ms': k['dominant_kernel_ms_per_item'], 'measured_on': 'rank 0'}
    except Exception as e:   # pylint: disable=broad-except
      roofline = {'error': f'{type(e).__name__}: {str(e)[:200]}'}
  return {
      'metric': 'non-uniform pts/s, batched 2D type-1 512^2 x 256 items, M=1e6 each, tol=1e-6 (set_points + execute), '
                'batch sharded over the GPUs',
      'value': round(value, 2), 'unit': 'Mpts/s', 'n_gpus': world, 'steps': args.steps,
      'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 4), 'higher_is_better': True,
      'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
      'roofline': roofline,
      'config': {
          'workload': f'BASELINE configs[4]: batched 2D type-1, 512x512 modes, batch={items}, M=1e6 per item, '
                      f'per-item points, tol=1e-6, complex64; contiguous blocks of the batch per rank '
                      f'(shard_bounds), each rank one tfft.nufft call per step over its block; no data-path '
                      f'collective. config.one_gpu_whole_job_Mpts_s is this workload on one GPU, '
                      f'config.equal_work_efficiency = value / (N x that).',
          'items': items, 'items_per_rank': nloc, 'points_per_item': m, 'grid': C5_GRID,
          'rccl_world_size': measured_world, 'backend': args.dist_backend if dist is not None else None,
          **extras,
      },
  }


if __name__ == '__main__':
  main()
