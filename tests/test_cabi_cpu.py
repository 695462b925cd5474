"""CPU-side checks of the C ABI (no GPU, no compute calls): the library loads,
exports every symbol include/nufft_hip.h declares, its structs have the layout
the Python binding assumes, the host-only parameter rules give the sizes of
SURVEY.md section 8, and validation errors carry the reference's messages."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import PKG, ROOT
from tensorflow_nufft import _lib


def _describe(rank, dims, tol, prec, ttype=1, ntransf=1, **opts):
  lib = _lib.lib()
  o = _lib.OptionsStruct()
  lib.nufft_hip_default_options(ctypes.byref(o))
  for k, v in opts.items():
    setattr(o, k, v)
  info = _lib.PlanInfo()
  err = ctypes.create_string_buffer(512)
  d = (ctypes.c_int64 * 3)(*(list(dims) + [1] * (3 - rank)))
  rc = lib.nufft_hip_plan_describe(ttype, rank, d, -1, ntransf, float(tol), prec, ctypes.byref(o),
                                   ctypes.byref(info), err, 512)
  return rc, err.value.decode(), info


def test_library_exports_every_declared_symbol():
  header = open(os.path.join(ROOT, 'include', 'nufft_hip.h')).read()
  declared = set(re.findall(r'\b(nufft_hip_[a-z_0-9]+)\s*\(', header))
  assert len(declared) >= 20
  handle = ctypes.CDLL(_lib.LIB_PATH)
  missing = [n for n in sorted(declared) if not hasattr(handle, n)]
  assert not missing, missing
  assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
  assert _lib.lib().nufft_hip_abi_version() == 4


def _source_digest():
  import hashlib
  import subprocess
  csrc = os.path.join(ROOT, 'tensorflow-nufft_amd', 'csrc')
  # the Makefile's own list, in its order
  r = subprocess.run(['make', '-s', '-C', csrc, '--eval', 'print-src: ; @echo $(SRC)', 'print-src'], capture_output=True, text=True)
  assert r.returncode == 0, r.stderr
  h = hashlib.sha256()
  for name in r.stdout.split():
    h.update(open(os.path.join(csrc, name), 'rb').read())
  return h.hexdigest()[:16]


def test_shipped_library_is_a_product_build_of_this_tree():
  # (r05 verdict / advisor) The kernels carry experiment switches -- knock-out builds that drop the atomics, the staging
  # reads, the result stores ("wrong results, timing only"), shape overrides, the phase log. csrc/nufft_experiment.h
  # refuses them without -DNUFFT_EXPERIMENT_BUILD, and nufft_hip_build_info() reports what a binary was built with:
  # the library the tests and the bench load must report none, the header's ABI version, and the digest of the
  # sources in this tree (a stale build fails here, not in a confusing parity test).
  info = dict(kv.split('=', 1) for kv in _lib.lib().nufft_hip_build_info().decode().split(';'))
  assert info['experiment'] == 'none', info
  assert info['abi'] == '4' and info['arch'] == 'gfx950', info
  assert info['source'] == _source_digest(), (info, 'libnufft_hip.so is older than csrc/: run make -C tensorflow-nufft_amd/csrc')


def test_experiment_macros_need_the_experiment_flag(tmp_path):
  # a stray -DNUFFT_GROUP_EXP=1 (or any other switch of nufft_experiment.h) must not compile into a product build,
  # and an experiment build says so
  import subprocess
  csrc = os.path.join(ROOT, 'tensorflow-nufft_amd', 'csrc')
  base = ['/opt/rocm/bin/hipcc', '-std=c++17', '-I' + os.path.join(ROOT, 'include'), '-I' + csrc, '-x', 'c++', '-fsyntax-only',
          os.path.join(csrc, 'nufft_build_info.cpp')]
  for macro in ('NUFFT_GROUP_EXP=1', 'NUFFT_DENSE_EXP=2', 'NUFFT_INTERP_EXP=8', 'NUFFT_GROUP_NW=8', 'NUFFT_DENSE_NW=8', 'NUFFT_PATCH_NW=16',
                'NUFFT_PATCH_MINW=2', 'NUFFT_STACK_ROWS=2', 'NUFFT_BOUND_THREADS=256', 'NUFFT_FX_BOUND_LIMIT=1e9', 'NUFFT_HIP_NO_PRELOAD',
                'NUFFT_HIP_PHASE_LOG', 'NUFFT_GROUP_STAGE=16', 'NUFFT_MIX_SHAPE_ENV'):
    r = subprocess.run(base + ['-D' + macro], capture_output=True, text=True)
    assert r.returncode != 0 and 'NUFFT_EXPERIMENT_BUILD' in r.stderr, (macro, r.stderr[-300:])
  exe = str(tmp_path / 'info')
  src = str(tmp_path / 'main.cpp')
  open(src, 'w').write('#include <cstdio>\nextern "C" const char* nufft_hip_build_info(void);\nint main() { std::puts(nufft_hip_build_info()); }\n')
  r = subprocess.run(['g++', '-std=c++17', '-I' + os.path.join(ROOT, 'include'), '-I' + csrc, '-DNUFFT_EXPERIMENT_BUILD', '-DNUFFT_GROUP_EXP=1',
                      '-DNUFFT_HIP_PHASE_LOG', os.path.join(csrc, 'nufft_build_info.cpp'), src, '-o', exe], capture_output=True, text=True)
  assert r.returncode == 0, r.stderr
  out = subprocess.run([exe], capture_output=True, text=True).stdout
  assert 'experiment=yes,NUFFT_GROUP_EXP=1,NUFFT_HIP_PHASE_LOG' in out, out


def test_default_options_and_struct_layout():
  o = _lib.OptionsStruct()
  _lib.lib().nufft_hip_default_options(ctypes.byref(o))
  assert o.points_range == 1 and o.max_batch_size == 0 and o.check_points_range == 0
  assert ctypes.sizeof(_lib.OptionsStruct) == 88
  # plan_describe echoes fields through the C struct: checks PlanInfo's layout
  rc, _, i = _describe(2, [1024, 1024], 1e-6, 4, ntransf=3)
  assert rc == 0 and (i.type, i.rank, i.precision, i.iflag, i.ntransf) == (1, 2, 4, -1, 3)
  assert list(i.grid_dims) == [1024, 1024, 1]


def test_tuning_bits_and_op_counts_are_validated():
  # (advisor, r03) the experiment surface is public ABI: unknown bits, OFF | ON pairs and negative counts are refused
  T = _lib.TUNE
  ok, _, _ = _describe(2, [64, 64], 1e-6, 4, tuning=T['GROUP_ON'] | T['NO_FUSED'] | T['SORT2_OFF'])
  assert ok == 0
  rc, err, _ = _describe(2, [64, 64], 1e-6, 4, tuning=1 << 30)
  assert rc == _lib.INVALID_ARGUMENT and 'unknown options.tuning bits' in err, (rc, err)
  # the table in _lib.py and NUFFT_HIP_TUNE_ALL in the header move together: every named bit is accepted alone,
  # the first unnamed one is refused
  for name, bit in T.items():
    rc, err, _ = _describe(3, [32, 32, 32], 1e-4, 4, tuning=bit)
    assert rc == 0, (name, err)
  rc, err, _ = _describe(2, [64, 64], 1e-6, 4, tuning=max(T.values()) << 1)
  assert rc == _lib.INVALID_ARGUMENT and 'unknown options.tuning bits' in err, (rc, err)
  for a, b in (('GROUP_OFF', 'GROUP_ON'), ('SPARSE_OFF', 'SPARSE_ON'), ('CELLSORT_OFF', 'CELLSORT_ON'),
               ('CELLSORT3D_OFF', 'CELLSORT3D_ON'), ('JOINT_OFF', 'JOINT_ON'), ('STAGED_OFF', 'STAGED_ON'),
               ('SORT2_OFF', 'SORT2_ON')):
    rc, err, _ = _describe(3, [32, 32, 32], 1e-4, 4, tuning=T[a] | T[b])
    assert rc == _lib.INVALID_ARGUMENT and 'OFF / ON pair' in err, (a, b, rc, err)
  for name in ('op_group', 'op_lanes'):
    rc, err, _ = _describe(2, [64, 64], 1e-6, 4, **{name: -1})
    assert rc == _lib.INVALID_ARGUMENT and 'must be >= 0' in err, (name, rc, err)


@pytest.mark.parametrize('rank,dims,tol,prec,w,nf,method', [
    (2, [1024, 1024], float(np.float32(1e-6)), 4, 8, [2048, 2048, 1], 2),   # BASELINE config 2/3
    (3, [256, 256, 256], float(np.float32(1e-4)), 4, 6, [512, 512, 512], 2),  # config 4
    (2, [512, 512], float(np.float32(1e-6)), 4, 8, [1024, 1024, 1], 2),     # config 5
    (1, [4096], float(np.float32(1e-6)), 8, 8, [8192, 1, 1], 1),            # config 1 (fp64)
    (2, [6, 8], 1e-6, 4, 8, [16, 16, 1], 1),                                # fine grid >= 2 w
])
def test_parameter_rules(rank, dims, tol, prec, w, nf, method):
  rc, err, i = _describe(rank, dims, tol, prec)
  assert rc == 0, err
  assert i.kernel_width == w and list(i.fine_dims) == nf and i.upsampling_factor == 2.0
  assert i.spread_method == method
  assert abs(i.beta - 2.30 * w) < 1e-12
  for d in range(rank):
    assert i.num_tiles[d] * i.tile_dims[d] >= i.fine_dims[d]


def test_kernel_width_rule_safe_side():
  # w = ceil(log10(10 / tol)) with exact powers of ten rounded UP
  for tol, w in ((1e-1, 3), (1e-2, 4), (1e-3, 5), (1e-4, 6), (1e-5, 7), (1e-6, 8), (2e-6, 7),
                 (1e-9, 11), (1e-12, 14), (1e-14, 16), (1e-16, 16)):
    rc, _, i = _describe(2, [64, 64], tol, 8)
    assert rc == 0 and i.kernel_width == w, (tol, i.kernel_width)
  # float clamps tol at 6e-8 (nufft_plan.h:84-89)
  assert _describe(2, [64, 64], 1e-12, 4)[2].kernel_width == 9


def test_batch_size_rule():
  # the reference's rule is min(ntransf, 8) (nufft_plan.cu.cc:1923-1928); here the 8 grows for small fine
  # grids (2^23 fine cells per batch), and stays for large ones
  assert _describe(2, [64, 64], 1e-6, 4, ntransf=20)[2].batch_size == 20
  assert _describe(2, [1024, 1024], 1e-6, 4, ntransf=20)[2].batch_size == 8
  assert _describe(2, [512, 512], 1e-6, 4, ntransf=100)[2].batch_size == 8
  assert _describe(2, [256, 256], 1e-6, 4, ntransf=100)[2].batch_size == 32
  assert _describe(2, [64, 64], 1e-6, 4, ntransf=3)[2].batch_size == 3
  assert _describe(2, [64, 64], 1e-6, 4, ntransf=20, max_batch_size=2)[2].batch_size == 2


def test_tile_rules_of_the_r02_kernels():
  # widths 9..16 (nufft_wide.hip): 32 x 32 tiles in 2-D, 16 x 8 x 4 up to w = 12, 8 x 8 x 4 above, 8 x 8 x 2 at
  # w = 16; grids smaller than a tile fall back to the thread-per-point kernels (method 1)
  for rank, dims, tol, prec, tile, method in (
      (2, [1024, 1024], 1e-9, 8, [32, 32, 1], 2), (2, [64, 64], 1e-12, 8, [32, 32, 1], 2),
      (3, [128, 128, 128], 1e-9, 8, [16, 8, 4], 2), (3, [64, 64, 64], 1e-12, 8, [8, 8, 4], 2),
      (3, [64, 64, 64], 1e-14, 8, [8, 8, 2], 2), (3, [128, 128, 128], 1e-12, 4, [16, 8, 4], 2)):   # float clamps to w = 9
    rc, err, i = _describe(rank, dims, tol, prec)
    assert rc == 0, err
    assert list(i.tile_dims) == tile and i.spread_method == method, (dims, tol, list(i.tile_dims), i.spread_method)
  rc, _, i = _describe(2, [8, 8], 1e-12, 8)            # fine grid 28 x 28 < 32: generic kernels
  assert rc == 0 and i.spread_method == 1
  # 3-D float w = 8: tiles of depth 8 (one fp64 plane per launch), double keeps depth 4
  assert list(_describe(3, [128, 128, 128], 1e-6, 4)[2].tile_dims) == [16, 16, 8]
  assert list(_describe(3, [128, 128, 128], 1e-6, 8)[2].tile_dims) == [16, 16, 4]
  # 2-D float type 2: 64 x 64 tiles from 2^21 fine cells on, 32 x 32 below; type 1 always 32 x 32
  assert list(_describe(2, [1024, 1024], 1e-6, 4, ttype=2)[2].tile_dims) == [64, 64, 1]
  assert list(_describe(2, [256, 256], 1e-6, 4, ttype=2)[2].tile_dims) == [32, 32, 1]
  assert list(_describe(2, [1024, 1024], 1e-6, 4, ttype=1)[2].tile_dims) == [32, 32, 1]
  # explicit tile sizes switch the specialised geometries off
  rc, _, i = _describe(2, [1024, 1024], 1e-9, 8, tile_dims=(ctypes.c_int32 * 3)(16, 16, 1))
  assert rc == 0 and i.spread_method == 1 and list(i.tile_dims) == [16, 16, 1]


def test_create_argument_errors():
  assert _describe(4, [8, 8, 8], 1e-6, 4)[0] == _lib.UNIMPLEMENTED
  rc, err, _ = _describe(2, [64, 64], 1e-6, 4, ttype=3)
  assert rc == _lib.UNIMPLEMENTED and 'type-3' in err
  rc, err, _ = _describe(2, [64, 64], 1e-6, 4, ntransf=0)
  assert rc == _lib.INVALID_ARGUMENT and 'num_transforms must be >= 1' in err
  rc, err, _ = _describe(2, [14, 64], 1e-6, 4, spread_only=1)
  assert rc == _lib.INVALID_ARGUMENT and 'Invalid grid dimension size: 14' in err
  rc, err, _ = _describe(2, [64, 64], 1e-6, 4, upsampling_factor=0.5)
  assert rc == _lib.INVALID_ARGUMENT and 'upsampling_factor must be > 1.0' in err


def _op_shape(op, ttype, prec, source_shape, points_shape, grid_shape=()):
  d = _lib.OpDesc()
  d.op_type, d.transform_type, d.fft_direction, d.precision, d.tol = op, ttype, -1, prec, 1e-6
  _lib.lib().nufft_hip_default_options(ctypes.byref(d.options))
  d.source_ndim, d.points_ndim, d.grid_shape_len = len(source_shape), len(points_shape), len(grid_shape)
  for i, s in enumerate(source_shape):
    d.source_shape[i] = s
  for i, s in enumerate(points_shape):
    d.points_shape[i] = s
  for i, s in enumerate(grid_shape[:3]):
    d.grid_shape[i] = s
  nd = ctypes.c_int32()
  shp = (ctypes.c_int64 * 12)()
  err = ctypes.create_string_buffer(512)
  rc = _lib.lib().nufft_hip_op_shape(ctypes.byref(d), ctypes.byref(nd), shp, err, 512)
  return rc, err.value.decode(), [shp[i] for i in range(nd.value)]


def test_op_shape_inference_and_errors():
  # nufft_ops_test.py:667-725 (static shapes) and :438-503 (error messages)
  assert _op_shape(0, 1, 4, [100], [100, 2], [8, 6])[2] == [8, 6]
  assert _op_shape(0, 2, 4, [8, 6], [100, 2])[2] == [100]
  assert _op_shape(0, 1, 4, [2, 4, 100], [1, 100, 2], [24, 24])[2] == [2, 4, 24, 24]
  assert _op_shape(0, 1, 4, [1, 100], [2, 4, 100, 2], [24, 24])[2] == [2, 4, 24, 24]
  assert _op_shape(0, 2, 4, [5, 1, 8, 8, 8], [3, 50, 3])[2] == [5, 3, 50]
  assert _op_shape(1, 2, 8, [3, 64, 96], [100, 2])[2] == [3, 100]        # Interp
  assert _op_shape(2, 1, 8, [100], [100, 3], [4, 8, 6])[2] == [4, 8, 6]  # Spread
  rc, err, _ = _op_shape(0, 1, 4, [100], [100, 2], [8])
  assert rc == _lib.INVALID_ARGUMENT and 'grid_shape must have length 2' in err
  rc, err, _ = _op_shape(0, 1, 4, [99], [100, 2], [8, 8])
  assert rc == _lib.INVALID_ARGUMENT and 'must have equal samples dimensions' in err
  rc, err, _ = _op_shape(0, 1, 4, [100], [100, 4], [8, 8, 8])
  assert rc == _lib.INVALID_ARGUMENT and 'Dimension must be 1, 2 or 3' in err
  rc, err, _ = _op_shape(0, 1, 4, [3, 100], [2, 100, 2], [8, 8])
  assert rc == _lib.INVALID_ARGUMENT and 'Incompatible shapes: [3,100] vs. [2,100,2]' in err
  rc, err, _ = _op_shape(0, 2, 4, [8], [100, 2])
  assert rc == _lib.INVALID_ARGUMENT and 'must have rank of at least 2' in err


def test_product_package_does_not_import_the_oracle():
  # the shipped path must never route through oracle/ (or any CPU fallback)
  pkg = os.path.join(ROOT, 'tensorflow-nufft_amd')
  for dirpath, _, files in os.walk(pkg):
    for f in files:
      if f.endswith(('.py', '.cpp', '.hip', '.h', '.cc')):
        text = open(os.path.join(dirpath, f), errors='replace').read()
        assert 'import oracle' not in text and 'from oracle' not in text, f
        assert 'nufft_oracle' not in text, f


def test_python_surface_fails_loudly_without_gpu():
  import torch
  if torch.cuda.device_count() > 0:
    pytest.skip('GPU present')
  import tensorflow_nufft as tfft
  with pytest.raises(RuntimeError, match='no CPU kernels'):
    tfft.nufft(np.zeros(4, np.complex64), np.zeros((4, 1), np.float32), grid_shape=[8], transform_type='type_1')
  with pytest.raises(ValueError, match='grid_shape must be provided for type-1 transforms'):
    tfft.nufft(np.zeros(4, np.complex64), np.zeros((4, 1), np.float32), transform_type='type_1')


def test_header_is_plain_c_and_links(tmp_path):
  # include/nufft_hip.h must be usable from C (the reference-side binding could be cgo/JNI/ctypes):
  # compile a C program against it with gcc -std=c99 -pedantic, link libnufft_hip.so, run a host-only call
  import shutil
  import subprocess
  if shutil.which('gcc') is None:
    pytest.skip('gcc not available')
  src = tmp_path / 'use_abi.c'
  src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "nufft_hip.h"
int main(void) {
  nufft_hip_options o;
  nufft_hip_plan_info info;
  char err[256];
  int64_t dims[3] = {1024, 1024, 1};
  int rc;
  nufft_hip_default_options(&o);
  if (nufft_hip_abi_version() != NUFFT_HIP_ABI_VERSION) return 2;
  rc = nufft_hip_plan_describe(NUFFT_HIP_TYPE_1, 2, dims, NUFFT_HIP_FORWARD, 1, 1e-6, NUFFT_HIP_F32, &o, &info,
                               err, sizeof err);
  if (rc != NUFFT_HIP_OK) { printf("%s\n", err); return 3; }
  printf("%d %d %d\n", (int)info.kernel_width, (int)info.fine_dims[0], (int)info.spread_method);
  return 0;
}
''')
  exe = tmp_path / 'use_abi'
  libdir = os.path.dirname(_lib.LIB_PATH)
  cmd = ['gcc', '-std=c99', '-pedantic', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), str(src),
         '-o', str(exe), '-L', libdir, '-lnufft_hip', '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib']
  r = subprocess.run(cmd, capture_output=True, text=True)
  assert r.returncode == 0, r.stderr
  r = subprocess.run([str(exe)], capture_output=True, text=True)
  assert r.returncode == 0, (r.stdout, r.stderr)
  assert r.stdout.split() == ['8', '2048', '2']


def test_stage_list_matches_the_header():
  # the Python stage names index the arrays nufft_hip_plan_get_timing fills
  import re
  from conftest import ROOT
  from tensorflow_nufft import _lib
  hdr = open(os.path.join(ROOT, 'include', 'nufft_hip.h')).read()
  n = int(re.search(r'#define NUFFT_HIP_NUM_STAGES (\d+)', hdr).group(1))
  assert n == len(_lib.STAGES)
  assert _lib.STAGES[-1] == 'sort_cell' and _lib.STAGES[4] == 'spread'


def test_tf_glue_type_checks_against_the_api_stub():
  # TensorFlow is absent from this image, so the TF glue cannot be built; it can be TYPE-CHECKED against
  # tests/tf_api_stub (declarations only, NOT TensorFlow): syntax, types, and every call into the C ABI.
  import subprocess
  src = os.path.join(ROOT, 'tensorflow-nufft_amd', 'csrc', 'tf_glue', 'nufft_tf_ops.cc')
  cmd = ['g++', '-std=c++17', '-fsyntax-only', '-Wall', '-Werror', '-Wno-unused-function',
         '-I', os.path.join(ROOT, 'tests', 'tf_api_stub'), '-I', os.path.join(ROOT, 'include'), src]
  r = subprocess.run(cmd, capture_output=True, text=True)
  assert r.returncode == 0, r.stderr[-3000:]
  # the check has teeth: a wrong argument to the C ABI is refused
  bad = open(src).read().replace('nufft_hip_op_shape(&d, &ndim, shape, err, sizeof(err))', 'nufft_hip_op_shape(&d, shape, &ndim, err, sizeof(err))')
  assert bad != open(src).read()
  r = subprocess.run(cmd[:-1] + ['-x', 'c++', '-'], input=bad, capture_output=True, text=True)
  assert r.returncode != 0
  # and the glue stays plumbing: no arithmetic on tensor contents, no parsing
  text = open(src).read()
  assert len(text.splitlines()) <= 210 and 'ReadVarint' not in text and 'ParseOptions' not in text


def test_tf_glue_shape_function_runs_on_the_reference_shape_cases(tmp_path):
  # The one piece of logic that lives only in the TF glue -- the graph-time shape function (BaseShapeFn /
  # NUFFTShapeFn, reference cc/ops/nufft_ops.cc:27-103) -- EXECUTED through a behavioural stand-in for
  # InferenceContext (tests/tf_api_stub/README.md; still not TensorFlow) on the reference tests' shape grid
  # (nufft_ops_test.py:87-98: three grids x three source batch shapes x four points batch shapes x both types),
  # the Interp / Spread cases, partially unknown shapes, and the two error messages.
  import itertools
  import subprocess
  exe = str(tmp_path / 'tf_shape_fn_driver')
  libdir = os.path.dirname(_lib.LIB_PATH)
  r = subprocess.run(['g++', '-std=c++17', '-Wall', '-Werror', '-Wno-unused-function', '-I', os.path.join(ROOT, 'tests', 'tf_api_stub'),
                      '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'tf_shape_fn_driver.cc'), '-o', exe,
                      '-L', libdir, '-lnufft_hip', '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib'],
                     capture_output=True, text=True)
  assert r.returncode == 0, r.stderr[-3000:]

  def fmt(shape):
    return ','.join('?' if d is None else str(d) for d in shape) if len(shape) else '-'

  cases, want = [], []
  for grid, sb, pb, ttype in itertools.product([[8], [6, 8], [4, 8, 6]], [[], [2, 4], [4]], [[], [2, 1], [1, 4], [4]],
                                               ['type_1', 'type_2']):
    m = int(np.prod(grid))
    batch = list(np.broadcast_shapes(tuple(sb), tuple(pb)))
    src = sb + ([m] if ttype == 'type_1' else grid)
    pts = pb + [m, len(grid)]
    cases.append(f'NUFFT {ttype} {fmt(src)} {fmt(pts)} v:{fmt(grid)}')
    want.append('OK [' + fmt(batch + (grid if ttype == 'type_1' else [m])).replace('-', '') + ']')
  extra = [
      ('Interp - 128,128 16384,2 none', 'OK [16384]'),                   # nufft_ops_test.py:224-252
      ('Spread - 4096 4096,2 v:64,64', 'OK [64,64]'),                    # :255-284
      ('Interp - 5,64,96 5,6144,2 none', 'OK [5,6144]'),                 # :287-316 (batch of sources)
      ('Spread - 5,6144 5,6144,2 v:64,96', 'OK [5,64,96]'),              # :319-348
      ('NUFFT type_1 2,4,576 1,576,2 v:24,24', 'OK [2,4,24,24]'),        # :351-417 (broadcast either way)
      ('NUFFT type_2 1,24,24 2,4,576,2 v:24,24', 'OK [2,4,576]'),
      # unknown pieces: the static shape is as specific as the inputs allow
      ('NUFFT type_2 ? ? r:?', 'OK ?'),                                  # rank of the transform unknown
      ('NUFFT type_1 ?,100 ?,100,2 r:2', 'OK [?,?,?]'),                  # grid_shape not constant: two unknown grid dims
      ('NUFFT type_1 7,? ?,3 v:4,?,6', 'OK [7,4,?,6]'),
      ('NUFFT type_2 ?,8,8 3,?,2 v:8,8', 'OK [3,?]'),
      ('NUFFT type_2 ? 5,40,3 r:3', 'OK ?'),                             # source of unknown rank: batch unknown
      ('Interp - 4,1,8,8 3,50,2 none', 'OK [4,3,50]'),
      # errors, with the reference's / TensorFlow's messages
      ('NUFFT type_2 8,8 64,4 v:8,8', 'ERR Dimension must be 1, 2 or 3, but is 4'),
      ('Spread - 10 10,4 v:2,2', 'ERR Dimension must be 1, 2 or 3, but is 4'),
      ('NUFFT type_1 3,20 3,21,2 v:4,4', 'ERR Dimensions must be equal, but are 21 and 20'),
      ('NUFFT type_1 2,48 3,48,2 v:6,8', 'ERR Dimensions must be equal, but are 2 and 3'),   # batch shapes do not broadcast
      ('NUFFT type_1 48 48,2 v:4,4,3', 'ERR Shape must be rank 2 but is rank 3'),            # grid_shape vs points
      ('NUFFT type_3 3 3,1 r:1', "ERR transform_type attr must be 'type_1' or 'type_2', but is type_3"),
  ]
  cases += [c for c, _ in extra]
  want += [w for _, w in extra]
  r = subprocess.run([exe], input='\n'.join(cases) + '\n', capture_output=True, text=True)
  assert r.returncode == 0, r.stderr[-2000:]
  got = r.stdout.splitlines()
  assert len(got) == len(want) == 72 + len(extra)
  for c, g, w in zip(cases, got, want):
    assert g == w, (c, g, w)



def test_no_kernel_spills_registers():
  # r04 verdict: 17 instantiations of the sort kernels (general coordinate fold, double precision) and the cell sorts carried
  # 24-172 bytes of scratch per lane. Every __global__ of csrc/ is checked here (hipcc cross-compiles without a GPU;
  # -Rpass-analysis=kernel-resource-usage): no scratch anywhere, except the one general-fold staged scatter that keeps
  # 8 bytes under the 128 VGPRs a 1024-thread workgroup can have.
  import concurrent.futures
  import re
  import subprocess
  src = os.path.join(PKG, 'csrc')
  files = [f for f in sorted(os.listdir(src)) if f.endswith('.hip')]
  allowed = {'scatter_staged_kernel<float, 0, true, 1024, false, false>': 8}

  def analyse(f):
    r = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'), '-I' + src,
                        '--offload-arch=gfx950', '-munsafe-fp-atomics', '--cuda-device-only', '-Rpass-analysis=kernel-resource-usage',
                        '-c', os.path.join(src, f), '-o', os.devnull], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    out = []
    for b in re.split(r'remark: [^\n]*Function Name: ', r.stderr)[1:]:
      name = b.split('\n')[0].strip()
      m = re.search(r'ScratchSize \[bytes/lane\]: (\d+)', b)
      out.append((name, int(m.group(1)) if m else -1))
    return out

  with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
    results = [x for rows in ex.map(analyse, files) for x in rows]
  assert len(results) > 200          # every instantiation was seen
  names = subprocess.run(['c++filt'], input='\n'.join(n for n, _ in results), capture_output=True, text=True).stdout.splitlines()
  bad = []
  for (_, scratch), dn in zip(results, names):
    limit = max([v for k, v in allowed.items() if k in dn] or [0])
    if scratch > limit or scratch < 0:
      bad.append((dn[:150], scratch))
  assert not bad, bad
