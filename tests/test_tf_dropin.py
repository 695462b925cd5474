"""The drop-in as TensorFlow sees it (-m gpu; SKIPS where TensorFlow is absent -- it is in this repository's image).

tools/build_tf_glue.sh builds `_nufft_ops.so` from csrc/tf_glue/nufft_tf_ops.cc against an installed TensorFlow-ROCm
and points NUFFT_TF_OPS_SO at it; this file then does what the reference's Python layer does
(tensorflow_nufft/python/ops/nufft_ops.py:26-27 `tf.load_op_library`, :118-123 the `nufft` op call with the
serialized options proto) and checks the three registered ops against the committed NUDFT golden vectors, the
reference's shape errors and one gradient through `tf.GradientTape` with the reference's registered-gradient algebra
(nufft_ops.py:126-232) restated on the raw ops. It turns "never compiled against TensorFlow" 
from a caveat into a recipe; until somebody runs it, the caveat stands.
"""
import os

import numpy as np
import pytest

from conftest import rel_l2

pytestmark = pytest.mark.gpu

tf = pytest.importorskip('tensorflow')


@pytest.fixture(scope='module')
def ops():
  path = os.environ.get('NUFFT_TF_OPS_SO')
  if not path or not os.path.exists(path):
    pytest.skip('NUFFT_TF_OPS_SO is not set: build the glue with tools/build_tf_glue.sh')
  assert tf.config.list_physical_devices('GPU'), 'the ops are registered for DEVICE_GPU only'
  return tf.load_op_library(path)


def _options_bytes():
  # an empty Options message: every field at its proto3 default (points_range STRICT), which is what
  # nufft_options.Options().to_proto().SerializeToString() gives for the defaults except points_range
  return b''


def _cases(golden, fname):
  g = golden(fname)
  for name in g['names']:
    name = str(name)
    _, tt1, tt2, fd = name.rsplit('_', 3)
    grid = [int(v) for v in name.split('_')[0][1:].split('x')]
    yield name, grid, f'{tt1}_{tt2}', fd, g[name + '_points'], g[name + '_source'], g[name + '_target']


def test_registered_ops_and_attrs(ops):
  for n in ('nufft', 'interp', 'spread'):
    assert hasattr(ops, n), n


@pytest.mark.parametrize('fname', ['nudft_cases.npz', 'nudft_mid.npz'])
def test_nufft_op_matches_the_golden_nudft(ops, golden, fname):
  for name, grid, tt, fd, pts, src, target in _cases(golden, fname):
    with tf.device('/GPU:0'):
      gs = tf.constant(grid if tt == 'type_1' else [], dtype=tf.int32)
      out = ops.nufft(tf.constant(src), tf.constant(pts), gs, transform_type=tt, fft_direction=fd, tol=1e-6,
                      options=_options_bytes()).numpy()
    assert out.shape == target.shape
    assert rel_l2(out, target) < 1e-6, (name, rel_l2(out, target))


def test_interp_and_spread_ops_are_adjoint(ops):
  rng = np.random.default_rng(5)
  grid, M = [24, 32], 5000
  pts = rng.uniform(-np.pi, np.pi, (M, 2)).astype(np.float32)
  f = (rng.standard_normal(grid) + 1j * rng.standard_normal(grid)).astype(np.complex64)
  c = (rng.standard_normal(M) + 1j * rng.standard_normal(M)).astype(np.complex64)
  with tf.device('/GPU:0'):
    fi = ops.interp(tf.constant(f), tf.constant(pts), tol=1e-6, options=_options_bytes()).numpy()
    cs = ops.spread(tf.constant(c), tf.constant(pts), tf.constant(grid, dtype=tf.int32), tol=1e-6, options=_options_bytes()).numpy()
  lhs, rhs = np.vdot(c, fi), np.vdot(cs, f)
  assert abs(lhs - rhs) / abs(lhs) < 1e-5, (lhs, rhs)


def test_shape_errors_are_the_reference_ones(ops):
  pts = tf.zeros([10, 2], tf.float32)
  with pytest.raises((tf.errors.InvalidArgumentError, ValueError)):
    ops.nufft(tf.zeros([10], tf.complex64), pts, tf.constant([8, 8, 8], tf.int32), transform_type='type_1',
              fft_direction='forward', tol=1e-6, options=_options_bytes())   # rank of grid_shape != points.shape[-1]


def test_gradient_through_the_raw_ops(ops):
  # d/d source of sum |A source|^2 for a type-2 transform: 2 A^H A source -- the adjoint is the type-1 op with the
  # opposite sign (the algebra of the reference's registered gradient, nufft_ops.py:126-160)
  rng = np.random.default_rng(6)
  grid, M = [16, 20], 3000
  pts = tf.constant(rng.uniform(-np.pi, np.pi, (M, 2)).astype(np.float32))
  f = tf.constant((rng.standard_normal(grid) + 1j * rng.standard_normal(grid)).astype(np.complex64))
  with tf.device('/GPU:0'):
    empty = tf.constant([], tf.int32)
    y = ops.nufft(f, pts, empty, transform_type='type_2', fft_direction='forward', tol=1e-6, options=_options_bytes())
    want = 2.0 * ops.nufft(y, pts, tf.constant(grid, tf.int32), transform_type='type_1', fft_direction='backward', tol=1e-6,
                           options=_options_bytes())
    # finite differences along one random direction
    d = tf.constant((rng.standard_normal(grid) + 1j * rng.standard_normal(grid)).astype(np.complex64))
    eps = 1e-2
    def loss(x):
      return tf.reduce_sum(tf.abs(ops.nufft(x, pts, empty, transform_type='type_2', fft_direction='forward', tol=1e-6,
                                            options=_options_bytes())) ** 2)
    num = (loss(f + eps * d) - loss(f - eps * d)) / (2 * eps)
  ana = tf.math.real(tf.reduce_sum(tf.math.conj(want) * d))
  assert abs(float(num) - float(ana)) / abs(float(ana)) < 1e-3
