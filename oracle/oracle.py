"""ctypes front end of the CPU oracle + the dense NUDFT definition.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg. The product package never imports this module.

Conventions follow the reference's Python surface
(tensorflow_nufft/python/ops/nufft_ops.py:34-123): `points` is [M, rank] with
the last axis ordered like the grid dimensions, grids are C-contiguous in
`grid_shape` order; the reversal to FINUFFT's x-fastest order done by the
reference at cc/kernels/nufft_kernels.cc:276-303,347-352 happens here.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

RANGE = {'strict': 0, 'extended': 1, 'infinite': 2}


class OracleOpts(ctypes.Structure):
  _fields_ = [('type', ctypes.c_int32), ('rank', ctypes.c_int32),
              ('N', ctypes.c_int64 * 3), ('iflag', ctypes.c_int32),
              ('ntransf', ctypes.c_int32), ('tol', ctypes.c_double),
              ('sigma', ctypes.c_double), ('w', ctypes.c_int32),
              ('spread_only', ctypes.c_int32),
              ('points_range', ctypes.c_int32),
              ('kerevalmeth', ctypes.c_int32), ('nthreads', ctypes.c_int32)]


class OracleInfo(ctypes.Structure):
  _fields_ = [('sigma', ctypes.c_double), ('w', ctypes.c_int32),
              ('ncoef', ctypes.c_int32), ('beta', ctypes.c_double),
              ('nf', ctypes.c_int64 * 3)]


def build(force=False):
  """Compiles the oracle (and the reference pieces when /root/reference exists)."""
  so = os.path.join(_HERE, '_build', 'libnufft_oracle.so')
  srcs = [os.path.join(_HERE, f) for f in ('nufft_oracle.c', 'nufft_oracle_impl.h')]
  stale = (not os.path.exists(so) or
           any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs))
  if force or stale:
    subprocess.run(['make', '-C', _HERE, '_build/libnufft_oracle.so'],
                   check=True, capture_output=True)
  ref_so = os.path.join(_HERE, '_ref', 'libnufft_ref.so')
  if os.path.isdir('/root/reference/tensorflow_nufft') and (
      force or not os.path.exists(ref_so)):
    subprocess.run(['make', '-C', _HERE, 'ref'], check=True, capture_output=True)
  return so


def lib():
  global _LIB
  if _LIB is None:
    _LIB = ctypes.CDLL(build())
    _LIB.oracle_next_smooth_even.restype = ctypes.c_int64
    _LIB.oracle_next_smooth_even.argtypes = [ctypes.c_int64]
  return _LIB


def ref_lib():
  """The standalone-compilable reference pieces (None when not built)."""
  global _REF
  if _REF is None:
    p = os.path.join(_HERE, '_ref', 'libnufft_ref.so')
    if not os.path.exists(p):
      return None
    _REF = ctypes.CDLL(p)
  return _REF


def _ptr(a):
  return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


_DEFAULT_THREADS = None


def default_threads():
  """OpenMP threads when the caller passes nthreads = 0: the CPUs this process may really use -- its affinity mask
  capped by the cgroup CPU quota (the GPU boxes show 256 hardware threads under a quota of 16 CPUs; a team of 256
  throttled threads makes the oracle several times slower and its timing erratic)."""
  global _DEFAULT_THREADS
  if _DEFAULT_THREADS is None:
    try:
      n = len(os.sched_getaffinity(0))
    except AttributeError:
      n = os.cpu_count() or 1
    try:
      q, p = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
      if q != 'max':
        n = max(1, min(n, int(int(q) / int(p) + 0.5)))
    except (OSError, ValueError):
      try:
        q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if q > 0:
          n = max(1, min(n, int(q / p + 0.5)))
      except (OSError, ValueError):
        pass
    _DEFAULT_THREADS = n
  return _DEFAULT_THREADS


def _mk_opts(transform_type, rank, grid_shape, fft_direction, ntransf, tol,
             sigma, w, spread_only, points_range, kerevalmeth, nthreads):
  if not nthreads:
    nthreads = default_threads()
  o = OracleOpts()
  o.type = 1 if transform_type == 'type_1' else 2
  o.rank = rank
  for d in range(3):
    o.N[d] = int(grid_shape[rank - 1 - d]) if d < rank else 1
  o.iflag = -1 if fft_direction == 'forward' else 1
  o.ntransf = ntransf
  o.tol = float(tol)
  o.sigma = float(sigma)
  o.w = int(w)
  o.spread_only = int(spread_only)
  o.points_range = RANGE[points_range]
  o.kerevalmeth = int(kerevalmeth)
  o.nthreads = int(nthreads)
  return o


def query(rank, grid_shape, tol, precision='f32', sigma=0.0, w=0,
          spread_only=False):
  """Returns (sigma, w, beta, nf[rank] in grid_shape order) the rules select."""
  o = _mk_opts('type_1', rank, grid_shape, 'forward', 1, tol, sigma, w,
               spread_only, 'extended', 0, 1)
  info = OracleInfo()
  rc = lib().oracle_query(ctypes.byref(o), 4 if precision == 'f32' else 8,
                          ctypes.byref(info))
  if rc:
    raise ValueError(f'oracle_query failed with code {rc}')
  return info.sigma, info.w, info.beta, [info.nf[rank - 1 - d] for d in range(rank)]


def nufft(source, points, grid_shape=None, transform_type='type_2',
          fft_direction='forward', tol=1e-6, sigma=0.0, w=0,
          points_range='extended', kerevalmeth=0, nthreads=0, op='nufft',
          tol_as_f32_attr=True):
  """One points set, optional leading transform axis on `source`.

  type_1: source [T?, M] -> [T?, *grid_shape];  type_2: source [T?, *grid] -> [T?, M].
  dtype complex64 runs the float instantiation, complex128 the double one.
  op: 'nufft' | 'spread' | 'interp' (the latter two: no upsampling, scaled).
  tol_as_f32_attr: round tol through float32 first, as the reference op does
  (`tol: float` attr, cc/ops/nufft_ops.cc:214; cast at nufft_kernels.cc:361),
  which is what makes w = 8 at tol=1e-6 also in double precision.
  """
  if tol_as_f32_attr:
    tol = float(np.float32(tol))
  source = np.asarray(source)
  points = np.asarray(points)
  cdt = source.dtype
  if cdt not in (np.complex64, np.complex128):
    raise TypeError('source must be complex64 or complex128')
  rdt = np.float32 if cdt == np.complex64 else np.float64
  suf = '_f32' if cdt == np.complex64 else '_f64'
  M, rank = points.shape
  pts = np.ascontiguousarray(points.T[::-1].astype(rdt))  # row 0 = last coord
  if transform_type == 'type_1':
    grid_shape = [int(g) for g in grid_shape]
    batched = source.ndim == 2
    src = np.ascontiguousarray(source.reshape(-1, M))
  else:
    grid_shape = list(source.shape[-rank:])
    batched = source.ndim == rank + 1
    src = np.ascontiguousarray(source.reshape((-1,) + tuple(grid_shape)))
  T = src.shape[0]
  o = _mk_opts(transform_type, rank, grid_shape, fft_direction, T, tol, sigma,
               w, op != 'nufft', points_range, kerevalmeth, nthreads)
  info = OracleInfo()
  if transform_type == 'type_1':
    c = src
    f = np.zeros((T,) + tuple(grid_shape), dtype=cdt)
  else:
    f = src
    c = np.zeros((T, M), dtype=cdt)
  fn = getattr(lib(), ('oracle_nufft' if op == 'nufft' else 'oracle_spread_interp') + suf)
  rc = fn(ctypes.byref(o), ctypes.c_int64(M), _ptr(pts[0]),
          _ptr(pts[1]) if rank > 1 else None,
          _ptr(pts[2]) if rank > 2 else None, _ptr(c), _ptr(f),
          ctypes.byref(info))
  if rc:
    raise ValueError(f'oracle failed with code {rc}')
  out = f if transform_type == 'type_1' else c
  return out if batched else out[0]


def time_nufft(c, points, grid_shape, tol=1e-6, sigma=0.0, w=0, kerevalmeth=1, nthreads=0, repeats=5):
  """Seconds per call (list of `repeats` samples, after one warm-up call) of the C entry `oracle_nufft` alone for a
  type-1 transform: bin sort + spread + FFT + deconvolve on inputs that are ALREADY in the layout the C code takes
  (coordinate arrays, x first). `nufft()` above also transposes the [M, rank] points with numpy on one thread --
  ~50 ms at M = 1e7, which is not part of the algorithm being timed. For bench.py's cpu_baseline leg."""
  import time
  tol = float(np.float32(tol))
  c = np.ascontiguousarray(c)
  cdt = c.dtype
  rdt = np.float32 if cdt == np.complex64 else np.float64
  suf = '_f32' if cdt == np.complex64 else '_f64'
  M, rank = points.shape
  pts = np.ascontiguousarray(np.asarray(points).T[::-1].astype(rdt))
  grid_shape = [int(g) for g in grid_shape]
  o = _mk_opts('type_1', rank, grid_shape, 'forward', 1, tol, sigma, w, False, 'extended', kerevalmeth, nthreads)
  info = OracleInfo()
  f = np.zeros((1,) + tuple(grid_shape), dtype=cdt)
  fn = getattr(lib(), 'oracle_nufft' + suf)
  args = (ctypes.byref(o), ctypes.c_int64(M), _ptr(pts[0]), _ptr(pts[1]) if rank > 1 else None,
          _ptr(pts[2]) if rank > 2 else None, _ptr(c), _ptr(f), ctypes.byref(info))
  if fn(*args):
    raise ValueError('oracle_nufft failed')
  out = []
  for _ in range(repeats):
    t0 = time.perf_counter()
    fn(*args)
    out.append(time.perf_counter() - t0)
  return out


def nudft(source, points, grid_shape=None, transform_type='type_2',
          fft_direction='forward', chunk=4096):
  """Dense float64 NUDFT. Definition from the reference's own test oracle
  (python/ops/nufft_ops.py:235-321): modes k_d = -N_d/2 .. N_d/2-1 in grid
  order, exp(-i k.x) for 'forward'; type 1 is the transpose. Integer modes
  -(N//2).. are used for odd N as the C++ does (nufft_plan.cc:733-734).
  source: [M] (type 1) or grid (type 2); points [M, rank]."""
  source = np.asarray(source).astype(np.complex128)
  points = np.asarray(points).astype(np.float64)
  M, rank = points.shape
  shape = list(grid_shape) if transform_type == 'type_1' else list(source.shape)
  sign = -1.0 if fft_direction == 'forward' else 1.0
  kvecs = [np.arange(-(n // 2), -(n // 2) + n, dtype=np.float64) for n in shape]
  if transform_type == 'type_1':
    out = np.zeros(shape, dtype=np.complex128)
  else:
    out = np.zeros(M, dtype=np.complex128)
  for s in range(0, M, chunk):
    p = points[s:s + chunk]
    # separable phases: E_d[j, k_d] = exp(sign i k_d x_jd)
    E = [np.exp(1j * sign * np.outer(p[:, d], kvecs[d])) for d in range(rank)]
    if transform_type == 'type_1':
      c = source[s:s + chunk]
      if rank == 1:
        out += c @ E[0]
      elif rank == 2:
        out += np.einsum('j,ja,jb->ab', c, E[0], E[1], optimize=True)
      else:
        out += np.einsum('j,ja,jb,jc->abc', c, E[0], E[1], E[2], optimize=True)
    else:
      if rank == 1:
        out[s:s + chunk] = E[0] @ source
      elif rank == 2:
        out[s:s + chunk] = np.einsum('ab,ja,jb->j', source, E[0], E[1], optimize=True)
      else:
        out[s:s + chunk] = np.einsum('abc,ja,jb,jc->j', source, E[0], E[1], E[2], optimize=True)
  return out


def spread_stage(c, points, grid_shape, tol=1e-6, sigma=2.0, w=0, points_range='extended',
                 kerevalmeth=0, nthreads=0):
  """Fine grid after the type-1 spreading step (reference spreadSorted,
  nufft_plan.cc:1027-1132, with the unnormalised kernel exp(beta sqrt(1 - c x^2))):
  complex array of shape nf (grid order). c: [M] complex64/128, points: [M, rank]."""
  c = np.ascontiguousarray(c)
  cdt = c.dtype
  rdt = np.float32 if cdt == np.complex64 else np.float64
  suf = '_f32' if cdt == np.complex64 else '_f64'
  M, rank = points.shape
  pts = np.ascontiguousarray(np.asarray(points).T[::-1].astype(rdt))
  o = _mk_opts('type_1', rank, grid_shape, 'forward', 1, tol, sigma, w, False, points_range,
               kerevalmeth, nthreads)
  info = OracleInfo()
  rc = lib().oracle_query(ctypes.byref(o), 4 if cdt == np.complex64 else 8, ctypes.byref(info))
  if rc:
    raise ValueError(f'oracle_query failed with code {rc}')
  nf = (ctypes.c_int64 * 3)(*[info.nf[d] for d in range(3)])
  perm = np.zeros(max(M, 1), dtype=np.int32)
  getattr(lib(), 'oracle_binsort' + suf)(
      ctypes.c_int64(M), _ptr(pts[0]), _ptr(pts[1]) if rank > 1 else None,
      _ptr(pts[2]) if rank > 2 else None, rank, nf, RANGE[points_range], _ptr(perm), int(nthreads or default_threads()))
  shape = [int(info.nf[rank - 1 - d]) for d in range(rank)]
  fw = np.zeros(shape, dtype=cdt)
  rc = getattr(lib(), 'oracle_spread_stage' + suf)(
      ctypes.byref(o), _ptr(perm), ctypes.c_int64(M), _ptr(pts[0]),
      _ptr(pts[1]) if rank > 1 else None, _ptr(pts[2]) if rank > 2 else None, _ptr(c), _ptr(fw))
  if rc:
    raise ValueError(f'oracle_spread_stage failed with code {rc}')
  return fw, info


def time_spread(c, points, grid_shape, tol=1e-6, sigma=0.0, w=0, kerevalmeth=1, nthreads=0):
  """Seconds of the type-1 spreader alone (spreadSorted on pre-sorted points; the bin sort
  runs before the clock starts). For bench.py's cpu_baseline leg."""
  import time
  c = np.ascontiguousarray(c)
  cdt = c.dtype
  rdt = np.float32 if cdt == np.complex64 else np.float64
  suf = '_f32' if cdt == np.complex64 else '_f64'
  M, rank = points.shape
  pts = np.ascontiguousarray(np.asarray(points).T[::-1].astype(rdt))
  o = _mk_opts('type_1', rank, grid_shape, 'forward', 1, tol, sigma, w, False, 'extended',
               kerevalmeth, nthreads)
  info = OracleInfo()
  if lib().oracle_query(ctypes.byref(o), 4 if cdt == np.complex64 else 8, ctypes.byref(info)):
    raise ValueError('oracle_query failed')
  nf = (ctypes.c_int64 * 3)(*[info.nf[d] for d in range(3)])
  perm = np.zeros(max(M, 1), dtype=np.int32)
  args = (_ptr(pts[0]), _ptr(pts[1]) if rank > 1 else None, _ptr(pts[2]) if rank > 2 else None)
  getattr(lib(), 'oracle_binsort' + suf)(ctypes.c_int64(M), *args, rank, nf, RANGE['extended'], _ptr(perm),
                                         int(nthreads or default_threads()))
  fw = np.zeros([int(info.nf[rank - 1 - d]) for d in range(rank)], dtype=cdt)
  fn = getattr(lib(), 'oracle_spread_stage' + suf)
  t0 = time.perf_counter()
  rc = fn(ctypes.byref(o), _ptr(perm), ctypes.c_int64(M), *args, _ptr(c), _ptr(fw))
  dt = time.perf_counter() - t0
  if rc:
    raise ValueError(f'oracle_spread_stage failed with code {rc}')
  return dt


def fft(a, sign, nthreads=0):
  """In-place-semantics FFT of the oracle (returns a new array); a is complex, C order."""
  a = np.array(a, copy=True, order='C')
  suf = '_f32' if a.dtype == np.complex64 else '_f64'
  nf = (ctypes.c_int64 * 3)(*([int(n) for n in a.shape[::-1]] + [1] * (3 - a.ndim)))
  getattr(lib(), 'oracle_fft' + suf)(_ptr(a), nf, a.ndim, int(sign), int(nthreads or default_threads()))
  return a


def eval_kernel(x1, tol=1e-6, sigma=2.0, w=0, kerevalmeth=0, precision='f64', rank=1,
                grid_shape=(64,)):
  rdt = np.float32 if precision == 'f32' else np.float64
  x1 = np.ascontiguousarray(x1, dtype=rdt)
  o = _mk_opts('type_1', rank, grid_shape, 'forward', 1, tol, sigma, w, False,
               'extended', kerevalmeth, 1)
  info = OracleInfo()
  lib().oracle_query(ctypes.byref(o), x1.itemsize, ctypes.byref(info))
  out = np.zeros((x1.size, info.w), dtype=rdt)
  getattr(lib(), 'oracle_eval_kernel' + ('_f32' if precision == 'f32' else '_f64'))(
      ctypes.byref(o), x1.size, _ptr(x1), _ptr(out))
  return out


def fseries(nf, tol=1e-6, sigma=2.0, w=0, precision='f64'):
  o = _mk_opts('type_1', 1, (64,), 'forward', 1, tol, sigma, w, False, 'extended', 0, 1)
  out = np.zeros(nf // 2 + 1, dtype=np.float64)
  lib().oracle_fseries(ctypes.byref(o), 4 if precision == 'f32' else 8,
                       ctypes.c_int64(nf), _ptr(out))
  return out


def gauss_legendre(n):
  z = np.zeros(n)
  w = np.zeros(n)
  lib().oracle_gauss_legendre(n, _ptr(z), _ptr(w))
  return z, w
