import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
PKG = os.path.join(ROOT, 'tensorflow-nufft_amd')
for p in (ROOT, PKG):
  if p not in sys.path:
    sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
  config.addinivalue_line('markers', 'sizecap: GPU tests on fine grids of 2^30 cells (tens of GB; skip with -m "gpu and not sizecap")')


def pytest_collection_modifyitems(config, items):
  # GPU tests are selected with -m gpu; skip them when no device is visible.
  try:
    import torch
    have_gpu = torch.cuda.device_count() > 0
  except Exception:  # pragma: no cover
    have_gpu = False
  if have_gpu:
    return
  skip = pytest.mark.skip(reason='no GPU visible')
  for item in items:
    if 'gpu' in item.keywords:
      item.add_marker(skip)


@pytest.fixture(scope='session')
def golden():
  def load(name):
    return np.load(os.path.join(GOLDEN, name))
  return load


def rel_l2(a, b):
  a = np.asarray(a).astype(np.complex128).ravel()
  b = np.asarray(b).astype(np.complex128).ravel()
  return float(np.linalg.norm(a - b) / np.linalg.norm(b))
