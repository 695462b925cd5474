"""ctypes binding of libnufft_hip.so (C ABI: include/nufft_hip.h).

The shared library is the product; this module only declares its entry points.
There is no CPU fallback: if the library is missing the import fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libnufft_hip.so')

OK = 0
INVALID_ARGUMENT = 3
RESOURCE_EXHAUSTED = 8
UNIMPLEMENTED = 12
INTERNAL = 13

TYPE_1, TYPE_2 = 1, 2
FORWARD, BACKWARD = -1, 1
F32, F64 = 4, 8
OP_NUFFT, OP_INTERP, OP_SPREAD = 0, 1, 2
METHOD_AUTO, METHOD_TILE_GENERIC, METHOD_TILE_WAVE, METHOD_POINT_GLOBAL = 0, 1, 2, 3
# nufft_hip_options.tuning bits (include/nufft_hip.h)
TUNE = {name: 1 << bit for bit, name in enumerate((
    'NO_FUSED', 'GROUP_OFF', 'GROUP_ON', 'SPARSE_OFF', 'SPARSE_ON', 'CELLSORT_OFF', 'CELLSORT_ON', 'CELLSORT3D_OFF',
    'CELLSORT3D_ON', 'ROCFFT', 'NO_WIDE', 'NO_LINE', 'JOINT_OFF', 'JOINT_ON', 'STAGED_OFF', 'STAGED_ON',
    'SORT2_OFF', 'SORT2_ON', 'FXPATCH_OFF', 'QFOLD_OFF', 'STACK_OFF', 'STACK_ON', 'FBGROUP_OFF', 'MIXFFT_OFF', 'ISPLIT_OFF', 'ISPLIT_ON', 'DIRECT_OFF', 'DIRECT_ON'))}
STAGES = ('sort_count', 'sort_scan', 'sort_scatter', 'zero', 'spread', 'fft', 'deconvolve', 'interp', 'sort_cell')


class InvalidArgumentError(ValueError):
  """Counterpart of tf.errors.InvalidArgumentError raised by the reference op."""


class UnimplementedError(NotImplementedError):
  pass


class ResourceExhaustedError(MemoryError):
  pass


class InternalError(RuntimeError):
  pass


class OptionsStruct(ctypes.Structure):
  _fields_ = [('max_batch_size', ctypes.c_int32),
              ('points_range', ctypes.c_int32),
              ('check_points_range', ctypes.c_int32),
              ('fftw_planning_rigor', ctypes.c_int32),
              ('spread_only', ctypes.c_int32),
              ('kernel_width', ctypes.c_int32),
              ('upsampling_factor', ctypes.c_double),
              ('spread_method', ctypes.c_int32),
              ('max_subproblem_size', ctypes.c_int32),
              ('tile_dims', ctypes.c_int32 * 3),
              ('lds_accumulate', ctypes.c_int32),
              ('num_point_sets', ctypes.c_int32),
              ('tuning', ctypes.c_int32),
              ('op_group', ctypes.c_int32),
              ('op_lanes', ctypes.c_int32),
              ('reserved', ctypes.c_int32 * 3)]


class PlanInfo(ctypes.Structure):
  _fields_ = ([(n, ctypes.c_int32) for n in (
      'type', 'rank', 'precision', 'iflag', 'ntransf', 'batch_size',
      'kernel_width', 'ncoef', 'spread_method')] +
              [('upsampling_factor', ctypes.c_double), ('beta', ctypes.c_double),
               ('tol', ctypes.c_double),
               ('grid_dims', ctypes.c_int64 * 3), ('fine_dims', ctypes.c_int64 * 3),
               ('tile_dims', ctypes.c_int32 * 3), ('num_tiles', ctypes.c_int32 * 3),
               ('max_subproblem_size', ctypes.c_int32),
               ('num_points', ctypes.c_int64), ('workspace_bytes', ctypes.c_int64)])


ALLOC_FN = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p)
FREE_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_void_p)


class Allocator(ctypes.Structure):
  """nufft_hip_allocator: device-memory callbacks of the host framework."""
  _fields_ = [('alloc', ALLOC_FN), ('free', FREE_FN), ('user', ctypes.c_void_p)]


class OpDesc(ctypes.Structure):
  _fields_ = [('op_type', ctypes.c_int32), ('transform_type', ctypes.c_int32),
              ('fft_direction', ctypes.c_int32), ('precision', ctypes.c_int32),
              ('tol', ctypes.c_double), ('options', OptionsStruct),
              ('source_ndim', ctypes.c_int32), ('points_ndim', ctypes.c_int32),
              ('grid_shape_len', ctypes.c_int32),
              ('source_shape', ctypes.c_int64 * 12),
              ('points_shape', ctypes.c_int64 * 12),
              ('grid_shape', ctypes.c_int64 * 3)]


# Every symbol include/nufft_hip.h declares (tests check the export list).
SYMBOLS = {
    'nufft_hip_abi_version': (ctypes.c_int, []),
    'nufft_hip_build_info': (ctypes.c_char_p, []),
    'nufft_hip_default_options': (None, [ctypes.POINTER(OptionsStruct)]),
    'nufft_hip_plan_create': (ctypes.c_int, [
        ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int,
        ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_int, ctypes.c_double,
        ctypes.c_int, ctypes.POINTER(OptionsStruct), ctypes.c_void_p,
        ctypes.c_char_p, ctypes.c_size_t]),
    'nufft_hip_plan_create_ex': (ctypes.c_int, [
        ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int,
        ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_int, ctypes.c_double,
        ctypes.c_int, ctypes.POINTER(OptionsStruct), ctypes.c_void_p,
        ctypes.POINTER(Allocator), ctypes.c_char_p, ctypes.c_size_t]),
    'nufft_hip_plan_create_host': (ctypes.c_int, [
        ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int,
        ctypes.POINTER(ctypes.c_int64), ctypes.c_int, ctypes.c_int, ctypes.c_double,
        ctypes.c_int, ctypes.POINTER(OptionsStruct), ctypes.c_char_p, ctypes.c_size_t]),
    'nufft_hip_plan_release_workspace': (ctypes.c_int, [ctypes.c_void_p]),
    'nufft_hip_plan_set_allocator': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(Allocator)]),
    'nufft_hip_execute_with_points': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
    'nufft_hip_debug_stop_after': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'nufft_hip_debug_sort_path': (ctypes.c_int, [ctypes.c_void_p]),
    'nufft_hip_debug_sub_bounds': (ctypes.c_int64, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.c_int64]),
    'nufft_hip_debug_stacks': (ctypes.c_int64, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int32), ctypes.c_int64]),
    'nufft_hip_debug_stack_params': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]),
    'nufft_hip_debug_shader_clock_mhz': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]),
    'nufft_hip_set_points': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_int64]),
    'nufft_hip_execute': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'nufft_hip_spread': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'nufft_hip_interp': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    'nufft_hip_plan_get_info': (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(PlanInfo)]),
    'nufft_hip_plan_describe': (ctypes.c_int, [
        ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.c_int,
        ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.POINTER(OptionsStruct),
        ctypes.POINTER(PlanInfo), ctypes.c_char_p, ctypes.c_size_t]),
    'nufft_hip_plan_set_stream': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    'nufft_hip_plan_set_timing': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int]),
    'nufft_hip_plan_get_timing': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32), ctypes.c_int]),
    'nufft_hip_last_error': (ctypes.c_char_p, [ctypes.c_void_p]),
    'nufft_hip_plan_destroy': (ctypes.c_int, [ctypes.c_void_p]),
    'nufft_hip_debug_fine_grid': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64)]),
    'nufft_hip_debug_copy_fine_grid': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]),
    'nufft_hip_debug_fseries': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.c_int64]),
    'nufft_hip_debug_eval_kernel': (ctypes.c_int, [
        ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
        ctypes.POINTER(ctypes.c_double)]),
    'nufft_hip_options_from_proto': (ctypes.c_int, [
        ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(OptionsStruct)]),
    'nufft_hip_op_desc_from_attrs': (ctypes.c_int, [
        ctypes.POINTER(OpDesc), ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_double,
        ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]),
    'nufft_hip_op_shape': (ctypes.c_int, [
        ctypes.POINTER(OpDesc), ctypes.POINTER(ctypes.c_int32),
        ctypes.POINTER(ctypes.c_int64), ctypes.c_char_p, ctypes.c_size_t]),
    'nufft_hip_op_compute': (ctypes.c_int, [
        ctypes.POINTER(OpDesc), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]),
    'nufft_hip_op_compute_ex': (ctypes.c_int, [
        ctypes.POINTER(OpDesc), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
        ctypes.c_void_p, ctypes.POINTER(Allocator), ctypes.c_char_p, ctypes.c_size_t]),
    'nufft_hip_op_clear_cache': (None, []),
    'nufft_hip_op_set_cache_limit': (None, [ctypes.c_int64]),
    'nufft_hip_op_cache_bytes': (ctypes.c_int64, []),
}

_lib = None


def lib():
  """Loads libnufft_hip.so; raises if it has not been built (no fallback)."""
  global _lib
  if _lib is None:
    if not os.path.exists(LIB_PATH):
      raise ImportError(
          f'{LIB_PATH} not found: build it with `python __graft_entry__.py` or '
          '`make -C tensorflow-nufft_amd/csrc`. This package has no CPU fallback.')
    handle = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
      fn = getattr(handle, name)
      fn.restype = res
      fn.argtypes = args
    _lib = handle
  return _lib


def raise_for_status(code, message):
  if code == OK:
    return
  if isinstance(message, bytes):
    message = message.decode('utf-8', 'replace')
  if code == INVALID_ARGUMENT:
    raise InvalidArgumentError(message)
  if code == UNIMPLEMENTED:
    raise UnimplementedError(message)
  if code == RESOURCE_EXHAUSTED:
    raise ResourceExhaustedError(message)
  raise InternalError(message or f'nufft_hip error code {code}')
