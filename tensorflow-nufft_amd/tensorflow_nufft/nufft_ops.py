"""`nufft`, `nudft`, `interp`, `spread` -- the reference's Python surface
(tensorflow_nufft/python/ops/nufft_ops.py:30-321) over the MI355X library.

Tensors are `torch.Tensor`s living on a ROCm device (torch is used for device
memory and streams only); numpy arrays are accepted for convenience and are
moved to `cuda:0` and back. All computation happens in libnufft_hip.so through
its C ABI (`nufft_hip_op_shape` / `nufft_hip_op_compute`), which carries the
op-level host logic of the reference's TensorFlow kernel
(cc/kernels/nufft_kernels.cc:54-542). There is no CPU implementation here.
"""
import ctypes

import numpy as np
import torch

from tensorflow_nufft import _lib
from tensorflow_nufft import nufft_options

_TRANSFORM_TYPES = {'type_1': _lib.TYPE_1, 'type_2': _lib.TYPE_2}
_FFT_DIRECTIONS = {'forward': _lib.FORWARD, 'backward': _lib.BACKWARD}
_COMPLEX_TO_REAL = {torch.complex64: torch.float32, torch.complex128: torch.float64}


def _options_bytes(options):
  """The `options` attr: serialized Options proto, as the reference wrapper sends it
  (python/ops/nufft_ops.py:118-123: `options or Options()`, always serialized)."""
  if isinstance(options, (bytes, bytearray)):
    return bytes(options)
  if options is None:
    return _default_options_bytes()
  return options.to_proto().SerializeToString()


_DEFAULT_OPTIONS_BYTES = None
_DESC_CACHE = {}   # configuration -> (validated OpDesc, output shape): see _run_op


def _default_options_bytes():
  """Serialized default Options, built once (r06: constructing and serializing the default Options model cost ~25 us
  of host time on every call -- as much as the plan-level call itself on a small transform)."""
  global _DEFAULT_OPTIONS_BYTES
  if _DEFAULT_OPTIONS_BYTES is None:
    _DEFAULT_OPTIONS_BYTES = nufft_options.Options().to_proto().SerializeToString()
  return _DEFAULT_OPTIONS_BYTES


def _apply_internal_options(o, options):
  """Expert knobs (InternalOptions upstream, cc/kernels/nufft_options.h:92-162)."""
  extra = getattr(options, '_internal', None) or {}
  for k, v in extra.items():
    if k == 'tile_dims':
      for i, t in enumerate(v):
        o.tile_dims[i] = int(t)
    else:
      setattr(o, k, v)


def _options_struct(options):
  """Options -> proto bytes -> nufft_hip_options, decoded by the library like the op attr."""
  o = _lib.OptionsStruct()
  data = _options_bytes(options)
  rc = _lib.lib().nufft_hip_options_from_proto(data, len(data), ctypes.byref(o))
  _lib.raise_for_status(rc, 'Unable to parse options string.')
  _apply_internal_options(o, options)
  return o


def _to_device(x, name):
  """Returns (tensor on a GPU, was_numpy)."""
  if isinstance(x, torch.Tensor):
    if not x.is_cuda:
      if not torch.cuda.is_available():
        raise RuntimeError(
            f'`{name}` is a CPU tensor and no ROCm device is available; '
            'tensorflow_nufft (MI355X build) has no CPU kernels.')
      return x.cuda(), False
    return x, False
  if not torch.cuda.is_available():
    raise RuntimeError('No ROCm device available; tensorflow_nufft (MI355X build) has no CPU kernels.')
  return torch.as_tensor(np.asarray(x)).cuda(), True


def _run_op(op_type, source, points, grid_shape, transform_type, fft_direction, tol, options):
  source, np_in = _to_device(source, 'source')
  points, _ = _to_device(points, 'points')
  if points.device != source.device:
    points = points.to(source.device)
  # dtype checks: reference nufft_kernels.cc:58-67
  if source.dtype not in _COMPLEX_TO_REAL:
    raise _lib.InvalidArgumentError(
        f'Input `source` must have type complex64 or complex128 but got: {source.dtype}')
  if points.dtype != _COMPLEX_TO_REAL[source.dtype]:
    raise _lib.InvalidArgumentError(
        f'Input `points` must have type {_COMPLEX_TO_REAL[source.dtype]} but got: {points.dtype}')
  lib = _lib.lib()
  err = ctypes.create_string_buffer(1024)
  data = _options_bytes(options) if op_type == _lib.OP_NUFFT else b''
  # r06: the validated descriptor and the output shape of a configuration are kept (the two C calls that build them, the
  # numpy round trip of grid_shape and the struct fills were ~8 us of the ~33 us a small call costs on the host). The
  # key holds everything they depend on; a configuration that fails validation raises before anything is stored.
  key = None
  if grid_shape is None or isinstance(grid_shape, (list, tuple)):
    extra = getattr(options, '_internal', None) or {}
    try:
      key = (op_type, transform_type, fft_direction, float(tol), source.dtype, data, tuple(source.shape), tuple(points.shape),
             None if grid_shape is None else tuple(int(g) for g in grid_shape),
             tuple(sorted((k, tuple(v) if isinstance(v, (list, tuple)) else v) for k, v in extra.items())))
      hit = _DESC_CACHE.get(key)
    except TypeError:
      key, hit = None, None
    if hit is not None:
      desc, tshape_list = hit
      source = source.contiguous()
      points = points.contiguous()
      target = torch.empty(tshape_list, dtype=source.dtype, device=source.device)
      dev = source.device
      if dev.index is None or dev.index == torch.cuda.current_device():
        stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib.nufft_hip_op_compute(ctypes.byref(desc), source.data_ptr(), points.data_ptr(),
                                      target.data_ptr(), ctypes.c_void_p(stream), err, len(err))
      else:
        with torch.cuda.device(dev):
          stream = torch.cuda.current_stream(dev).cuda_stream
          rc = lib.nufft_hip_op_compute(ctypes.byref(desc), source.data_ptr(), points.data_ptr(),
                                        target.data_ptr(), ctypes.c_void_p(stream), err, len(err))
      _lib.raise_for_status(rc, err.value)
      return target.cpu().numpy() if np_in else target
  desc = _lib.OpDesc()
  rc = lib.nufft_hip_op_desc_from_attrs(
      ctypes.byref(desc), op_type, transform_type.encode(), fft_direction.encode(), float(tol),
      _lib.F32 if source.dtype == torch.complex64 else _lib.F64, data, len(data), err, len(err))
  _lib.raise_for_status(rc, err.value)
  _apply_internal_options(desc.options, options)
  if source.dim() > 12 or points.dim() > 12:
    raise _lib.InvalidArgumentError('too many dimensions')
  desc.source_ndim = source.dim()
  desc.points_ndim = points.dim()
  for i, s in enumerate(source.shape):
    desc.source_shape[i] = int(s)
  for i, s in enumerate(points.shape):
    desc.points_shape[i] = int(s)
  gs = [] if grid_shape is None else [int(g) for g in np.asarray(
      grid_shape.cpu() if isinstance(grid_shape, torch.Tensor) else grid_shape).reshape(-1)]
  if len(gs) > 3:
    gs_len = len(gs)
    gs = gs[:3]
  else:
    gs_len = len(gs)
  desc.grid_shape_len = gs_len
  for i, g in enumerate(gs):
    desc.grid_shape[i] = g
  ndim = ctypes.c_int32(0)
  tshape = (ctypes.c_int64 * 12)()
  rc = lib.nufft_hip_op_shape(ctypes.byref(desc), ctypes.byref(ndim), tshape, err, len(err))
  _lib.raise_for_status(rc, err.value)
  source = source.contiguous()
  points = points.contiguous()
  tshape_list = [tshape[i] for i in range(ndim.value)]
  if key is not None:
    if len(_DESC_CACHE) >= 256:
      _DESC_CACHE.clear()
    _DESC_CACHE[key] = (desc, tshape_list)
  target = torch.empty(tshape_list, dtype=source.dtype, device=source.device)
  with torch.cuda.device(source.device):
    stream = torch.cuda.current_stream(source.device).cuda_stream
    rc = lib.nufft_hip_op_compute(ctypes.byref(desc), source.data_ptr(), points.data_ptr(),
                                  target.data_ptr(), ctypes.c_void_p(stream), err, len(err))
  _lib.raise_for_status(rc, err.value)
  return target.cpu().numpy() if np_in else target


def _check_enums(transform_type, fft_direction):
  if transform_type not in _TRANSFORM_TYPES:
    raise ValueError(
        f"transform_type must be 'type_1' or 'type_2', but got: {transform_type}")
  if fft_direction not in _FFT_DIRECTIONS:
    raise ValueError(
        f"fft_direction must be 'forward' or 'backward', but got: {fft_direction}")


class _NufftFunction(torch.autograd.Function):
  """Gradient of `nufft`, as registered by the reference
  (`_nufft_grad`, python/ops/nufft_ops.py:126-232): d/dsource is the NUFFT of
  the opposite type and sign; d/dpoints is a type-2 NUFFT of `source * k` (or
  `grad * k`) times -/+ i, real part. Composed from the same op."""

  @staticmethod
  def forward(ctx, source, points, grid_shape, transform_type, fft_direction, tol, options):
    ctx.save_for_backward(source, points)
    ctx.cfg = (grid_shape, transform_type, fft_direction, tol, options)
    return _run_op(_lib.OP_NUFFT, source, points, grid_shape, transform_type,
                   fft_direction, tol, options)

  @staticmethod
  def backward(ctx, grad):
    source, points = ctx.saved_tensors
    grid_shape, transform_type, fft_direction, tol, options = ctx.cfg
    rank = points.shape[-1]
    if transform_type == 'type_2':
      grid_shape = list(source.shape[-rank:])
    grad_type = 'type_2' if transform_type == 'type_1' else 'type_1'
    grad_dir = 'forward' if fft_direction == 'backward' else 'backward'
    grad_source = grad_points = None
    # torch hands over the gradient in the conjugate convention of TF
    # (dL/d conj(z)); the algebra below follows the reference line by line.
    if ctx.needs_input_grad[0]:
      grad_source = _run_op(_lib.OP_NUFFT, grad, points, grid_shape, grad_type, grad_dir, tol, options)
      source_elem_rank = 1 if transform_type == 'type_1' else rank
      grad_source = _reduce_to_shape(grad_source, source.shape, source_elem_rank)
    if ctx.needs_input_grad[1]:
      rdt = points.dtype
      vecs = [torch.arange(-(n // 2), -(n // 2) + n, dtype=rdt, device=points.device) for n in grid_shape]
      grid_points = torch.stack(torch.meshgrid(*vecs, indexing='ij'), dim=0).to(source.dtype)
      imag_unit = torch.tensor(-1j if fft_direction == 'forward' else 1j, dtype=source.dtype,
                               device=points.device)
      cgrad = torch.conj(grad)
      if transform_type == 'type_2':
        gp = _run_op(_lib.OP_NUFFT, source.unsqueeze(-(rank + 1)) * grid_points,
                     points.unsqueeze(-3), None, 'type_2', fft_direction, tol, options)
        gp = gp * cgrad.unsqueeze(-2) * imag_unit
      else:
        gp = _run_op(_lib.OP_NUFFT, cgrad.unsqueeze(-(rank + 1)) * grid_points,
                     points.unsqueeze(-3), None, 'type_2', fft_direction, tol, options)
        gp = gp * source.unsqueeze(-2) * imag_unit
      # TF's real-input convention: the reference keeps Re of the product built
      # with conj(grad); torch expects the same real gradient for real inputs.
      gp = torch.real(gp).transpose(-1, -2)
      grad_points = _reduce_to_shape(gp, points.shape, 2)
    return grad_source, grad_points, None, None, None, None, None


def _reduce_to_shape(x, shape, elem_rank):
  """Sums the broadcast batch dimensions back (BroadcastGradientArgs upstream)."""
  shape = list(shape)
  while x.dim() > len(shape):
    x = x.sum(dim=0)
  batch_nd = len(shape) - elem_rank
  for i in range(batch_nd):
    if shape[i] == 1 and x.shape[i] != 1:
      x = x.sum(dim=i, keepdim=True)
  return x.reshape(shape)


def nufft(source, points, grid_shape=None, transform_type='type_2',
          fft_direction='forward', tol=1e-6, options=None):
  """Computes the non-uniform discrete Fourier transform via NUFFT.

  Same contract as the reference `tfft.nufft`
  (python/ops/nufft_ops.py:34-123):

  Args:
    source: complex64/complex128. `[...] + grid_shape` for `'type_2'`,
      `[..., M]` for `'type_1'`.
    points: float32/float64 `[..., M, N]`, N in {1, 2, 3}, radians/sample, last
      axis ordered like the grid dimensions; batch dims broadcast with `source`.
    grid_shape: required for `'type_1'`.
    transform_type: `'type_1'` (non-uniform to uniform) or `'type_2'`.
    fft_direction: `'forward'` (exp(-i k x)) or `'backward'`.
    tol: requested relative precision.
    options: `Options`.

  Returns:
    `[...] + grid_shape` (type 1) or `[..., M]` (type 2).
  """
  if grid_shape is None and transform_type == 'type_1':
    raise ValueError("grid_shape must be provided for type-1 transforms")
  _check_enums(transform_type, fft_direction)
  needs_grad = any(isinstance(t, torch.Tensor) and t.requires_grad for t in (source, points))
  if needs_grad and torch.is_grad_enabled():
    return _NufftFunction.apply(source, points, grid_shape, transform_type, fft_direction, tol, options)
  return _run_op(_lib.OP_NUFFT, source, points, grid_shape, transform_type, fft_direction, tol, options)


def interp(source, points, tol=1e-6):
  """Interpolates a regular grid at arbitrary points (spreading kernel only, no
  FFT / deconvolution). Reference op `Interp`, cc/ops/nufft_ops.cc:136-166."""
  return _run_op(_lib.OP_INTERP, source, points, None, 'type_2', 'forward', tol, None)


def spread(source, points, grid_shape, tol=1e-6):
  """Spreads arbitrary points onto a regular grid (no FFT / deconvolution).
  Reference op `Spread`, cc/ops/nufft_ops.cc:169-201."""
  return _run_op(_lib.OP_SPREAD, source, points, grid_shape, 'type_1', 'forward', tol, None)


def nudft(source, points, grid_shape=None, transform_type='type_2', fft_direction='forward'):
  """Dense non-uniform DFT (O(M N); for testing), like the reference's `nudft`
  (python/ops/nufft_ops.py:235-321): builds exp(-/+ i k.x) explicitly with torch
  on whatever device the inputs live on. Batch dimensions broadcast."""
  _check_enums(transform_type, fft_direction)
  np_in = not isinstance(source, torch.Tensor)
  source = torch.as_tensor(np.asarray(source)) if np_in else source
  points = torch.as_tensor(np.asarray(points)) if not isinstance(points, torch.Tensor) else points
  points = points.to(source.device)
  rank = points.shape[-1]
  if transform_type == 'type_1':
    if grid_shape is None:
      raise ValueError("grid_shape must be provided for type-1 transforms")
    grid_shape = [int(g) for g in grid_shape]
  else:
    grid_shape = list(source.shape[-rank:])
  sign = -1.0 if fft_direction == 'forward' else 1.0
  vecs = [torch.arange(-(n // 2), -(n // 2) + n, dtype=points.dtype, device=points.device)
          for n in grid_shape]
  kgrid = torch.stack(torch.meshgrid(*vecs, indexing='ij'), dim=0).reshape(rank, -1)
  phase = torch.matmul(points, kgrid)                       # [..., M, Ngrid]
  mat = torch.exp(torch.complex(torch.zeros_like(phase), sign * phase)).to(source.dtype)
  if transform_type == 'type_1':
    out = torch.matmul(source.unsqueeze(-2), mat).squeeze(-2)   # [..., Ngrid]
    out = out.reshape(list(out.shape[:-1]) + grid_shape)
  else:
    src = source.reshape(list(source.shape[:-rank]) + [-1])
    out = torch.matmul(mat, src.unsqueeze(-1)).squeeze(-1)      # [..., M]
  return out.cpu().numpy() if np_in else out
