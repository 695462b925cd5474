"""`Plan`: thin Python handle on the C-ABI plan (create / set_points / execute),
for callers that reuse one plan across many executions (benchmarks, iterative
reconstruction). Mirrors the reference's C++ `Plan` interface
(cc/kernels/nufft_plan.h:223-256)."""
import ctypes

import torch

from tensorflow_nufft import _lib
from tensorflow_nufft.nufft_ops import _options_struct


class Plan:
  """grid_shape is in array (TensorFlow) order; points are [M, rank] with the
  last axis ordered like grid_shape."""

  def __init__(self, transform_type, grid_shape, fft_direction='forward', num_transforms=1,
               tol=1e-6, dtype=torch.complex64, options=None, device=None, spread_only=False,
               host_only=False, allocator=None, **internal):
    """host_only: every parameter rule and table, no device state (works without a GPU;
    info / fseries / eval_kernel only). allocator: a `_lib.Allocator` for the workspace."""
    self._handle = None
    self.lib = _lib.lib()
    self.host_only = bool(host_only)
    self._allocator = allocator   # keep the callbacks alive
    self.rank = len(grid_shape)
    self.grid_shape = [int(g) for g in grid_shape]
    self.type = transform_type
    self.ntransf = int(num_transforms)
    self.cdtype = dtype
    self.rdtype = torch.float32 if dtype == torch.complex64 else torch.float64
    if host_only:
      self.device = None
    else:
      self.device = torch.device(device if device is not None else 'cuda')
      if self.device.index is None:
        self.device = torch.device('cuda', torch.cuda.current_device())
    o = _options_struct(options)
    o.spread_only = int(spread_only)
    self.nsets = max(1, int(internal.get('num_point_sets', 0) or 0))   # K point sets handled together
    for k, v in internal.items():
      if k == 'tile_dims':
        for i, t in enumerate(v):
          o.tile_dims[i] = int(t)
      else:
        setattr(o, k, v)
    dims = (ctypes.c_int64 * 3)(*([self.grid_shape[self.rank - 1 - d] for d in range(self.rank)] +
                                  [1] * (3 - self.rank)))
    err = ctypes.create_string_buffer(1024)
    h = ctypes.c_void_p()
    args = (1 if transform_type == 'type_1' else 2, self.rank, dims,
            -1 if fft_direction == 'forward' else 1, self.ntransf, float(tol),
            4 if dtype == torch.complex64 else 8, ctypes.byref(o))
    if host_only:
      rc = self.lib.nufft_hip_plan_create_host(ctypes.byref(h), *args, err, len(err))
    else:
      with torch.cuda.device(self.device):
        stream = torch.cuda.current_stream(self.device).cuda_stream
        rc = self.lib.nufft_hip_plan_create_ex(
            ctypes.byref(h), *args, ctypes.c_void_p(stream),
            ctypes.byref(allocator) if allocator is not None else None, err, len(err))
    _lib.raise_for_status(rc, err.value)
    self._handle = h
    self.M = 0
    self._points = None

  def _check(self, rc):
    if rc:
      _lib.raise_for_status(rc, self.lib.nufft_hip_last_error(self._handle))

  def info(self):
    i = _lib.PlanInfo()
    self._check(self.lib.nufft_hip_plan_get_info(self._handle, ctypes.byref(i)))
    return i

  def _check_points(self, points):
    points = points.to(self.device, self.rdtype).contiguous()
    if self.nsets > 1:
      assert points.dim() == 3 and points.shape[0] == self.nsets and points.shape[2] == self.rank, \
          'a plan with num_point_sets = K takes points [K, M, rank]'
    else:
      assert points.dim() == 2 and points.shape[1] == self.rank
    return points

  def _lead(self, source, elem_rank):
    lead = [self.nsets] if self.nsets > 1 else []
    if self.ntransf > 1 or source.dim() > len(lead) + elem_rank:
      lead = lead + [self.ntransf]
    return lead

  def set_points(self, points):
    points = self._check_points(points)
    self._points = points   # keep alive until the sort kernels have consumed it
    self.M = points.shape[-2]
    es = points.element_size()
    base = points.data_ptr()
    r = self.rank
    with torch.cuda.device(self.device):
      self._check(self.lib.nufft_hip_set_points(
          self._handle, self.M, base + (r - 1) * es,
          base + (r - 2) * es if r > 1 else None,
          base + (r - 3) * es if r > 2 else None, r))

  def execute(self, source, out=None):
    source = source.to(self.device, self.cdtype).contiguous()
    lead = self._lead(source, 1 if self.type == 'type_1' else self.rank)
    if self.type == 'type_1':
      if out is None:
        out = torch.empty(lead + self.grid_shape, dtype=self.cdtype, device=self.device)
      c, f = source, out
    else:
      if out is None:
        out = torch.empty(lead + [self.M], dtype=self.cdtype, device=self.device)
      c, f = out, source
    with torch.cuda.device(self.device):
      self._check(self.lib.nufft_hip_execute(self._handle, c.data_ptr(), f.data_ptr()))
    return out

  def execute_with_points(self, points, source, out=None):
    """set_points + execute as one call (what a NUFFT op invocation does); a type-1 plan
    with one transform sorts the strengths along with the points. The points are consumed."""
    points = self._check_points(points)
    source = source.to(self.device, self.cdtype).contiguous()
    self._points = points
    self.M = points.shape[-2]
    lead = self._lead(source, 1 if self.type == 'type_1' else self.rank)
    if self.type == 'type_1':
      if out is None:
        out = torch.empty(lead + self.grid_shape, dtype=self.cdtype, device=self.device)
      c, f = source, out
    else:
      if out is None:
        out = torch.empty(lead + [self.M], dtype=self.cdtype, device=self.device)
      c, f = out, source
    es = points.element_size()
    base = points.data_ptr()
    r = self.rank
    with torch.cuda.device(self.device):
      self._check(self.lib.nufft_hip_execute_with_points(
          self._handle, self.M, base + (r - 1) * es,
          base + (r - 2) * es if r > 1 else None,
          base + (r - 3) * es if r > 2 else None, r, c.data_ptr(), f.data_ptr()))
    return out

  def set_stream(self, stream):
    """Moves the plan to another stream (a torch.cuda.Stream or a raw hipStream_t value); the caller orders the
    streams against each other, as with any buffer used on two streams."""
    raw = stream.cuda_stream if hasattr(stream, 'cuda_stream') else int(stream)
    self._check(self.lib.nufft_hip_plan_set_stream(self._handle, ctypes.c_void_p(raw)))

  def sort_path(self):
    """Debug: the sort the last set_points ran (0 LDS histogram, 1 16-bit histogram + ranked scatter, 2 global
    counters, 3 two levels), -1 before any points are set."""
    return int(self.lib.nufft_hip_debug_sort_path(self._handle))

  def sub_bounds(self):
    """Debug: the count-filter bounds of a 3-D float w = 7 / 8 plan's subproblems (numpy float32; > 0 bound,
    < 0 left to the fp64 planes, 0 unused launch slot); empty for other plans."""
    import numpy as np
    n = int(self.lib.nufft_hip_debug_sub_bounds(self._handle, None, 0))
    if n < 0:
      self._check(1)
    out = np.zeros(max(n, 0), dtype=np.float32)
    if n > 0:
      self.lib.nufft_hip_debug_sub_bounds(self._handle, out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), n)
    return out

  def stacks(self):
    """Debug: the stacks of tiles a 3-D plan spreads over (int32 [n, 4]: column, z0 | nz << 16, point range of a
    piece or -1, -1); empty for plans that spread per subproblem. Float w = 7 / 8 plans: as many rows as
    sub_bounds() then has entries (one count-filter bound per stack); float w <= 6 plans and the double-precision /
    w = 9..16 plans (r06) cut stacks too but keep no per-stack bounds: sub_bounds() is empty there."""
    import numpy as np
    n = int(self.lib.nufft_hip_debug_stacks(self._handle, None, 0))
    if n < 0:
      self._check(1)
    out = np.zeros((max(n, 0), 4), dtype=np.int32)
    if n > 0:
      self.lib.nufft_hip_debug_stacks(self._handle, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), n)
    return out

  def stack_params(self, length=0, cap=0):
    """Debug: at most `length` tiles and `cap` points per stack from the next set_points on (0: the plan's rule)."""
    self._check(self.lib.nufft_hip_debug_stack_params(self._handle, int(length), int(cap)))

  def stop_after(self, stage):
    """Debug: execute returns after the named stage ('spread', 'fft', 'deconvolve'); None = run all."""
    self._check(self.lib.nufft_hip_debug_stop_after(
        self._handle, -1 if stage is None else _lib.STAGES.index(stage)))

  def fine_grid(self):
    """Debug: the plan's fine grid [batch, nf...] (array order) as a tensor view copy."""
    i = self.info()
    shape = [i.batch_size * self.nsets] + [int(i.fine_dims[self.rank - 1 - d]) for d in range(self.rank)]
    out = torch.empty(shape, dtype=self.cdtype, device=self.device)
    with torch.cuda.device(self.device):
      self._check(self.lib.nufft_hip_debug_copy_fine_grid(self._handle, out.data_ptr(), out.numel()))
    return out

  def fseries(self, dim):
    """The kernel's Fourier series phi-hat[0..nf/2] of grid dimension `dim` (array order), float64."""
    import numpy as np
    i = self.info()
    d = self.rank - 1 - dim
    n = int(i.fine_dims[d]) // 2 + 1
    out = np.zeros(n)
    self._check(self.lib.nufft_hip_debug_fseries(
        self._handle, d, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), n))
    return out

  def eval_kernel(self, x1):
    """Host evaluation of the plan's piecewise polynomial at offsets x1 in [-w/2, -w/2+1]: [n, w]."""
    import numpy as np
    x1 = np.ascontiguousarray(x1, dtype=np.float64)
    w = int(self.info().kernel_width)
    out = np.zeros((x1.size, w))
    self._check(self.lib.nufft_hip_debug_eval_kernel(
        self._handle, x1.size, x1.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
        out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
    return out

  def spread(self, c, out=None):
    c = c.to(self.device, self.cdtype).contiguous()
    lead = self._lead(c, 1)
    if out is None:
      out = torch.empty(lead + self.grid_shape, dtype=self.cdtype, device=self.device)
    with torch.cuda.device(self.device):
      self._check(self.lib.nufft_hip_spread(self._handle, c.data_ptr(), out.data_ptr()))
    return out

  def interp(self, f, out=None):
    f = f.to(self.device, self.cdtype).contiguous()
    lead = self._lead(f, self.rank)
    if out is None:
      out = torch.empty(lead + [self.M], dtype=self.cdtype, device=self.device)
    with torch.cuda.device(self.device):
      self._check(self.lib.nufft_hip_interp(self._handle, out.data_ptr(), f.data_ptr()))
    return out

  def set_timing(self, enable=True):
    """0/False off, 1/True every stage, 2 only the spread / interp kernel."""
    self._check(self.lib.nufft_hip_plan_set_timing(self._handle, int(enable)))

  def get_timing(self):
    """{stage: (total_ms, calls)} since the last call; synchronises the stream."""
    n = len(_lib.STAGES)
    ms = (ctypes.c_double * n)()
    calls = (ctypes.c_int32 * n)()
    self._check(self.lib.nufft_hip_plan_get_timing(self._handle, ms, calls, n))
    return {name: (ms[i], calls[i]) for i, name in enumerate(_lib.STAGES)}

  def close(self):
    if self._handle is not None:
      self.lib.nufft_hip_plan_destroy(self._handle)
      self._handle = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass
