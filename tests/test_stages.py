"""Stage-level parity: every stage of the HIP plan against the reference-built
fixtures and the CPU oracle, not only the end-to-end transform.

CPU part (no GPU; host-only plans, `nufft_hip_plan_create_host`):
  * the plan's piecewise-polynomial kernel vs the reference's generated Horner
    tables as compiled from /root/reference by oracle/Makefile -> oracle/_ref and
    committed as tests/golden/ref_horner.npz: the CPU tables
    (kernel_horner_sigma2.inc, `ker_s2_*`) and the GPU tables
    (kernel_horner_sigma2_gpu.inc, `ker_gpu_*`, nufft_plan.cu.cc:454-462);
  * the plan's kernel Fourier series vs the oracle's restatement of
    kernel_fseries_1d (nufft_util.cc:71-117) to 1e-13.
GPU part (-m gpu; `nufft_hip_debug_stop_after`):
  * fine grid after the type-1 spread vs the oracle's spreadSorted stage
    (nufft_plan.cc:1027-1132), 2-D and 3-D, every spread kernel family;
  * fine grid after the type-2 amplify step vs the definition
    (nufft_plan.cc:765-778) built from the oracle's Fourier series;
  * fine grid after the FFT vs the oracle FFT of the oracle spread.
The HIP kernel is normalised to phi(0) = 1 (the reference leaves exp(beta)),
so fine grids differ by exp(beta)^rank and Fourier series by exp(beta); both
cancel in the transform and are divided out here.
"""
import numpy as np
import pytest

from conftest import rel_l2
from oracle import oracle


def _host_plan(grid, tol=1e-6, w=0, dtype='c64', ttype='type_1'):
  import torch
  from tensorflow_nufft.plan import Plan
  return Plan(ttype, grid, tol=tol, dtype=torch.complex64 if dtype == 'c64' else torch.complex128,
              host_only=True, kernel_width=w)


# --------------------------------------------------------------- CPU: kernel

@pytest.mark.parametrize('w', range(4, 17))
@pytest.mark.parametrize('table', ['s2', 'gpu'])
def test_plan_kernel_matches_reference_tables(golden, w, table):
  g = golden('ref_horner.npz')
  key = f'ker_{table}_w{w}'
  if key not in g:
    pytest.skip(f'no {key} in the fixture')
  x1 = g[f'x1_w{w}']
  ref = g[key][:, :w]
  plan = _host_plan([64, 64], w=w, dtype='c128')
  ours = plan.eval_kernel(x1) * np.exp(plan.info().beta)   # undo the phi(0) = 1 normalisation
  plan.close()
  # the reference tables are fits whose error shrinks ~10x per unit of w (they sit below the
  # tolerance w serves, same bound as tests/test_oracle.py); ours has the same property
  scale = np.abs(ref).max()
  err = np.abs(ref[1:-1] - ours[1:-1]).max() / scale
  bound = max(10.0 ** (1.0 - w) * 3.0, 1e-12)
  assert err < bound, (w, table, err, bound)


@pytest.mark.parametrize('w', [2, 3, 5, 8, 12, 16])
def test_plan_kernel_matches_oracle_formula(w):
  x1 = np.linspace(-w / 2, -w / 2 + 1, 203)[1:-1]
  plan = _host_plan([64, 64], w=w, dtype='c128')
  ours = plan.eval_kernel(x1) * np.exp(plan.info().beta)
  plan.close()
  ref = oracle.eval_kernel(x1, w=w, kerevalmeth=0)
  # fit plateau ~5e-(w+1) of the peak (EXPERIMENTS.md section 1)
  assert np.abs(ours - ref).max() / np.abs(ref).max() < 10.0 ** (-w) * 30 + 1e-13


def test_plan_kernel_is_even():
  # the grouped 2-D spreader evaluates cell W-1-q as cell q at -z (nufft_kernels.hip): the
  # fitted table must have that symmetry to rounding
  for w in range(2, 9):
    plan = _host_plan([64, 64], w=w)
    x1 = np.linspace(-w / 2, -w / 2 + 1, 41)
    k = plan.eval_kernel(x1)
    plan.close()
    # z -> -z is x1 -> (1 - w) - x1, i.e. the reversed sample order
    assert np.abs(k - k[::-1, ::-1]).max() < 1e-13, w


# ---------------------------------------------------- CPU: Fourier series

@pytest.mark.parametrize('nf', [16, 2048, 8192])
@pytest.mark.parametrize('tol,w', [(1e-6, 8), (1e-4, 6), (1e-12, 14)])
def test_plan_fseries_matches_oracle(nf, tol, w):
  plan = _host_plan([nf // 2], tol=tol, dtype='c128')
  i = plan.info()
  assert i.kernel_width == w
  nf = int(i.fine_dims[0])   # = the request, except that a fine grid is never below 2 w cells
  ours = plan.fseries(0) * np.exp(i.beta)
  plan.close()
  ref = oracle.fseries(nf, tol=tol, sigma=2.0, w=w, precision='f64')
  assert ours.shape == ref.shape
  assert np.abs(ours - ref).max() / np.abs(ref).max() < 1e-13


# ------------------------------------------------------------ GPU: stages

def _dev(a):
  import torch
  return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _inputs(rank, M, seed):
  rng = np.random.default_rng(seed)
  pts = rng.uniform(-np.pi, np.pi, (M, rank)).astype(np.float32)
  c = (rng.uniform(-.5, .5, M) + 1j * rng.uniform(-.5, .5, M)).astype(np.complex64)
  return pts, c


SPREAD_CASES = [
    # grid, M, tol (-> w), extra plan options, what it exercises
    ([256, 256], 200_000, 1e-6, {}, 'cell-grouped 2-D kernel (dense: 0.76 pts / fine cell)'),
    ([256, 256], 20_000, 1e-6, {}, 'per-point 2-D w = 8 kernel'),
    ([256, 256], 20_000, 1e-4, {}, 'per-point 2-D kernel, w = 6'),
    ([256, 256], 1_000, 1e-6, {}, 'LDS-free kernel for sparse point sets'),
    ([256, 256], 50_000, 1e-6, {'spread_method': 1}, 'generic tile kernel'),
    ([256, 256], 50_000, 1e-6, {'spread_method': 3}, 'LDS-free kernel, forced'),
    ([48, 64, 40], 150_000, 1e-4, {}, '3-D fixed-point accumulation (w = 6)'),
    ([48, 64, 40], 150_000, 1e-4, {'lds_accumulate': 1}, '3-D fp64 planes (w = 6)'),
    ([32, 32, 32], 60_000, 1e-6, {}, '3-D w = 8 split re / im planes'),
    ([32, 32, 32], 300, 1e-6, {}, '3-D LDS-free kernel'),
    ([4096], 100_000, 1e-6, {}, '1-D'),
]


@pytest.mark.gpu
@pytest.mark.parametrize('grid,M,tol,opts,what', SPREAD_CASES, ids=[c[4] for c in SPREAD_CASES])
def test_fine_grid_after_spread_matches_oracle(grid, M, tol, opts, what):
  import torch
  from tensorflow_nufft.plan import Plan
  rank = len(grid)
  pts, c = _inputs(rank, M, 7)
  plan = Plan('type_1', grid, tol=tol, **opts)
  w = int(plan.info().kernel_width)
  beta = plan.info().beta
  plan.stop_after('spread')
  plan.set_points(_dev(pts))
  plan.execute(_dev(c))
  got = plan.fine_grid()[0].cpu().numpy().astype(np.complex128) * np.exp(beta * rank)
  # the same through the one-call entry (the dense 2-D case sorts the strengths into the records)
  plan.execute_with_points(_dev(pts), _dev(c))
  got2 = plan.fine_grid()[0].cpu().numpy().astype(np.complex128) * np.exp(beta * rank)
  plan.close()
  ref, info = oracle.spread_stage(c.astype(np.complex128), pts.astype(np.float64), grid, tol=tol, sigma=2.0, w=w)
  assert got.shape == ref.shape
  # float positions / kernel values: the transform's tolerance applies to the stage as well
  assert rel_l2(got, ref) < max(tol, 1e-6), (what, rel_l2(got, ref))
  assert rel_l2(got2, ref) < max(tol, 1e-6), (what, rel_l2(got2, ref))


@pytest.mark.gpu
@pytest.mark.parametrize('grid', [[96, 128], [24, 40, 32]])
def test_fine_grid_after_fft_matches_oracle(grid):
  from tensorflow_nufft.plan import Plan
  rank = len(grid)
  pts, c = _inputs(rank, 30_000, 8)
  plan = Plan('type_1', grid, fft_direction='backward', tol=1e-6)
  beta, w = plan.info().beta, int(plan.info().kernel_width)
  plan.stop_after('fft')
  plan.set_points(_dev(pts))
  plan.execute(_dev(c))
  got = plan.fine_grid()[0].cpu().numpy().astype(np.complex128) * np.exp(beta * rank)
  plan.close()
  fw, _ = oracle.spread_stage(c.astype(np.complex128), pts.astype(np.float64), grid, tol=1e-6, sigma=2.0, w=w)
  ref = oracle.fft(fw, +1)   # unnormalised, exponent sign = fft_direction (nufft_plan.cc:411-427)
  assert rel_l2(got, ref) < 1e-6, rel_l2(got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize('grid', [[96, 128], [24, 40, 32], [60, 50], [45, 64], [250], [20, 18, 25]])
def test_fused_fft_passes_equal_fft_then_deconvolve(grid):
  # The product path has no fine grid "after the FFT": its passes crop and deconvolve as they go (power-of-two
  # dimensions: fft_rotate_kernel; the other smooth sizes, r06: fft_mixed_kernel). The stage pair they replace is
  # the reference's FFT (nufft_plan.cc:411-427) followed by deconvolve_*d dir 1 (nufft_plan.cc:722-760):
  # f[k] = FFT(fw)[k mod nf] / prod_d phihat_d[|k_d|] -- built here from the oracle's spread stage, numpy's FFT and
  # the oracle's Fourier series.
  from tensorflow_nufft.plan import Plan
  rank = len(grid)
  pts, c = _inputs(rank, 30_000, 8)
  plan = Plan('type_1', grid, fft_direction='backward', tol=1e-6)
  i = plan.info()
  w = int(i.kernel_width)
  nf = [int(i.fine_dims[rank - 1 - d]) for d in range(rank)]
  plan.set_points(_dev(pts))
  got = plan.execute(_dev(c)).cpu().numpy().astype(np.complex128)
  plan.close()
  fw, _ = oracle.spread_stage(c.astype(np.complex128), pts.astype(np.float64), grid, tol=1e-6, sigma=2.0, w=w)
  F = oracle.fft(fw, +1)
  ks = [np.arange(-(n // 2), -(n // 2) + n) for n in grid]
  ph = [oracle.fseries(nf[d], tol=1e-6, sigma=2.0, w=w) for d in range(rank)]
  fac = [1.0 / ph[d][np.abs(ks[d])] for d in range(rank)]
  scale = fac[0]
  for d in range(1, rank):
    scale = np.multiply.outer(scale, fac[d])
  ref = F[np.ix_(*[ks[d] % nf[d] for d in range(rank)])] * scale
  assert rel_l2(got, ref) < 1e-6, rel_l2(got, ref)


@pytest.mark.gpu
@pytest.mark.parametrize('grid', [[10, 16], [9, 12], [6, 8, 10]])
def test_fine_grid_after_amplify_matches_definition(grid):
  # type-2 step 1 (reference deconvolve_*d with dir 2, nufft_plan.cc:765-778; GPU Amplify*
  # nufft_plan.cu.cc:383-435): fw zeroed, then fw[k mod nf] = f[k] / prod_d phihat_d[|k_d|]
  from tensorflow_nufft.plan import Plan
  rank = len(grid)
  rng = np.random.default_rng(9)
  f = (rng.standard_normal(grid) + 1j * rng.standard_normal(grid)).astype(np.complex64)
  pts, _ = _inputs(rank, 100, 9)
  plan = Plan('type_2', grid, tol=1e-6)
  i = plan.info()
  w, beta = int(i.kernel_width), i.beta
  nf = [int(i.fine_dims[rank - 1 - d]) for d in range(rank)]
  plan.stop_after('deconvolve')
  plan.set_points(_dev(pts))
  plan.execute(_dev(f))
  got = plan.fine_grid()[0].cpu().numpy().astype(np.complex128) / np.exp(beta * rank)
  plan.close()
  ref = np.zeros(nf, np.complex128)
  ks = [np.arange(-(n // 2), -(n // 2) + n) for n in grid]
  ph = [oracle.fseries(nf[d], tol=1e-6, sigma=2.0, w=w) for d in range(rank)]
  fac = [1.0 / ph[d][np.abs(ks[d])] for d in range(rank)]
  idx = np.ix_(*[ks[d] % nf[d] for d in range(rank)])
  scale = fac[0]
  for d in range(1, rank):
    scale = np.multiply.outer(scale, fac[d])
  ref[idx] = f.astype(np.complex128) * scale
  assert rel_l2(got, ref) < 5e-7, rel_l2(got, ref)
  # every fine cell outside the kept modes is exactly zero (the memset the reference does)
  mask = np.ones(nf, bool)
  mask[idx] = False
  assert np.all(got[mask] == 0)
