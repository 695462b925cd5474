"""One run at the reference's size cap (-m gpu; own marker `sizecap` so the soak / electric-fence scripts can skip it:
`-m "gpu and not sizecap"`).

The reference plan accepts fine grids of up to kMaxArraySize = 2e9 elements (nufft_plan.h:62, 843-848); 288 GB of HBM
is where such grids become usable. Here: fine grids of 2^30 cells -- 3-D 512^3 modes (1024^3 fine, 8.6 GB in complex64,
524 288 tiles in 4096 super-tiles: the two-level sort at its limit) and 2-D 16384^2 modes (32768^2 fine: rocFFT, > 10^6 tiles: the global-counter sort)
-- where the interleaved float index 2 * cell passes 2^31 in the last rows / planes. Nothing of this size fits an
oracle run; the checks are size independent: dense fp64 NUDFTs of all points on a few hundred modes taken from every
corner of the mode box (type 1), of all modes at a few hundred points planted next to the domain's ends and drawn
at random (type 2), adjointness <A c, f> = <c, A* f>, and linearity on the whole output.
"""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.sizecap]


def _rnd_c(shape, g):
  import torch
  return torch.complex(torch.rand(shape, generator=g, device='cuda') - .5, torch.rand(shape, generator=g, device='cuda') - .5)


def _plant(pts):
  """Points next to the ends of the domain (the last / first fine cells, wrapped stencils) and at its centre."""
  import torch
  rank = pts.shape[1]
  vals = [-np.pi, np.nextafter(np.float32(np.pi), np.float32(0)), -np.pi + 1e-4, np.pi - 1e-4, 0.0, 1e-7]
  k = 0
  for code in range(len(vals) ** rank):
    if k >= 96:
      break
    if code % 5 and rank == 3:   # (a fifth of the 216 corner combinations in 3-D)
      continue
    for d in range(rank):
      pts[k, d] = float(vals[(code // len(vals) ** d) % len(vals)])
    k += 1
  return k


def _mode_sets(n, device):
  import torch
  idx = torch.tensor([0, 1, n // 2 - 1, n // 2, n // 2 + 1, n - 2, n - 1], device=device)
  return idx, (idx - n // 2).to(torch.float64)


def test_3d_512_modes_on_a_1024_cubed_fine_grid():
  import torch
  import tensorflow_nufft as tfft
  N, M, tol = 512, 20_000_000, 1e-4
  grid = [N, N, N]
  g = torch.Generator(device='cuda').manual_seed(30)
  pts = (torch.rand((M, 3), generator=g, device='cuda') * 2 - 1) * np.pi
  nplant = _plant(pts)
  c = _rnd_c(M, g)
  plan = tfft.Plan('type_1', grid, 'forward', tol=tol)
  info = plan.info()
  assert [int(v) for v in info.fine_dims] == [1024, 1024, 1024]
  assert int(info.num_tiles[0]) * int(info.num_tiles[1]) * int(info.num_tiles[2]) == 524288
  plan.set_points(pts)
  assert plan.sort_path() == 3          # two levels: 4096 super-tiles of 64^3 cells (the staged kernel's 4096-destination form), then tiles
  Ac = plan.execute(c)
  # ---- type 1 against the definition on 7^3 modes from every corner, edge and the centre of the mode box
  idx, ks = _mode_sets(N, 'cuda')
  sub = torch.zeros((7, 7, 7), dtype=torch.complex128, device='cuda')
  for s in range(0, M, 2_000_000):
    p = pts[s:s + 2_000_000].to(torch.float64)
    t = c[s:s + 2_000_000].to(torch.complex128)[:, None] * torch.exp(-1j * p[:, 0:1] * ks)
    sub += torch.einsum('ja,jb,jc->abc', t, torch.exp(-1j * p[:, 1:2] * ks), torch.exp(-1j * p[:, 2:3] * ks))
  got = Ac[idx][:, idx][:, :, idx].to(torch.complex128)
  err = float(torch.linalg.norm(got - sub) / torch.linalg.norm(sub))
  assert err < tol, err
  worst = float((got - sub).abs().max() / sub.abs().max())
  assert worst < 5 * tol, worst          # (no single mode is off: an index overflow would put a whole row elsewhere)
  # ---- linearity on the whole output: A (2.5 c + c2) = 2.5 A c + A c2
  c2 = _rnd_c(M, g)
  Ac2 = plan.execute(c2)
  Amix = plan.execute(2.5 * c + c2)
  lin = float(torch.linalg.norm(Amix - (2.5 * Ac + Ac2)) / torch.linalg.norm(Amix))
  assert lin < tol, lin
  del Ac2, Amix, c2
  plan.close()
  # ---- type 2 on the same points: the definition at the planted and 100 random points, then adjointness
  f = _rnd_c(grid, g)
  plan2 = tfft.Plan('type_2', grid, 'backward', tol=tol)
  plan2.set_points(pts)
  AHf = plan2.execute(f)
  plan2.close()
  sel = torch.cat([torch.arange(nplant, device='cuda'), torch.randint(nplant, M, (100,), generator=g, device='cuda')])
  kk = torch.arange(-(N // 2), N - N // 2, device='cuda', dtype=torch.float64)
  p = pts[sel].to(torch.float64)
  e = [torch.exp(1j * p[:, d:d + 1] * kk) for d in range(3)]      # [n, N] each; array order: axis 0 pairs with pts[:, 0]
  want = torch.empty(sel.numel(), dtype=torch.complex128, device='cuda')
  f128 = f.to(torch.complex128)
  for i in range(0, sel.numel(), 16):
    t = torch.einsum('ja,abc->jbc', e[0][i:i + 16], f128)
    want[i:i + 16] = torch.einsum('jbc,jb,jc->j', t, e[1][i:i + 16], e[2][i:i + 16])
  del f128, t
  got2 = AHf[sel].to(torch.complex128)
  err2 = float(torch.linalg.norm(got2 - want) / torch.linalg.norm(want))
  assert err2 < tol, err2
  assert float((got2[:nplant] - want[:nplant]).abs().max() / want.abs().max()) < 5 * tol   # the points at the domain's ends
  lhs = torch.vdot(f.reshape(-1).to(torch.complex128), Ac.reshape(-1).to(torch.complex128))
  rhs = torch.vdot(AHf.to(torch.complex128), c.to(torch.complex128))
  assert abs(lhs - rhs) / abs(lhs) < 1e-5, (lhs, rhs)
  del Ac, AHf, f
  torch.cuda.empty_cache()


def test_2d_16384_modes_on_a_32768_squared_fine_grid():
  import torch
  import tensorflow_nufft as tfft
  N, M, tol = 16384, 20_000_000, 1e-6
  grid = [N, N]
  g = torch.Generator(device='cuda').manual_seed(31)
  pts = (torch.rand((M, 2), generator=g, device='cuda') * 2 - 1) * np.pi
  nplant = _plant(pts)
  c = _rnd_c(M, g)
  plan = tfft.Plan('type_1', grid, 'forward', tol=tol)
  info = plan.info()
  assert [int(info.fine_dims[d]) for d in range(2)] == [32768, 32768]
  plan.set_points(pts)
  path = plan.sort_path()
  assert path in (1, 2), path            # more than 16384 tiles: the ranked or the global-counter scatter
  Ac = plan.execute(c)
  plan.close()
  idx, ks = _mode_sets(N, 'cuda')
  sub = torch.zeros((7, 7), dtype=torch.complex128, device='cuda')
  for s in range(0, M, 4_000_000):
    p = pts[s:s + 4_000_000].to(torch.float64)
    t = c[s:s + 4_000_000].to(torch.complex128)[:, None] * torch.exp(-1j * p[:, 0:1] * ks)
    sub += torch.einsum('ja,jb->ab', t, torch.exp(-1j * p[:, 1:2] * ks))
  got = Ac[idx][:, idx].to(torch.complex128)
  err = float(torch.linalg.norm(got - sub) / torch.linalg.norm(sub))
  assert err < 2 * tol, err               # (7^2 modes of 2.7e8: the sample's own scatter around the 2.1e-7 of the whole output)
  assert float((got - sub).abs().max() / sub.abs().max()) < 10 * tol
  f = _rnd_c(grid, g)
  plan2 = tfft.Plan('type_2', grid, 'backward', tol=tol)
  plan2.set_points(pts)
  AHf = plan2.execute(f)
  plan2.close()
  sel = torch.cat([torch.arange(nplant, device='cuda'), torch.randint(nplant, M, (64,), generator=g, device='cuda')])
  kk = torch.arange(-(N // 2), N - N // 2, device='cuda', dtype=torch.float64)
  p = pts[sel].to(torch.float64)
  want = torch.empty(sel.numel(), dtype=torch.complex128, device='cuda')
  e1 = torch.exp(1j * p[:, 1:2] * kk)
  for r0 in range(0, N, 2048):            # row blocks of f: [n, 2048] x [2048, N] in complex128
    e0 = torch.exp(1j * p[:, 0:1] * kk[r0:r0 + 2048])
    part = (e0 @ f[r0:r0 + 2048].to(torch.complex128) * e1).sum(dim=1)
    want = part if r0 == 0 else want + part
  got2 = AHf[sel].to(torch.complex128)
  err2 = float(torch.linalg.norm(got2 - want) / torch.linalg.norm(want))
  assert err2 < 2 * tol, err2
  assert float((got2[:nplant] - want[:nplant]).abs().max() / want.abs().max()) < 10 * tol
  lhs = torch.vdot(f.reshape(-1).to(torch.complex128), Ac.reshape(-1).to(torch.complex128))
  rhs = torch.vdot(AHf.to(torch.complex128), c.to(torch.complex128))
  assert abs(lhs - rhs) / abs(lhs) < 1e-5, (lhs, rhs)
  del Ac, AHf, f
  torch.cuda.empty_cache()
