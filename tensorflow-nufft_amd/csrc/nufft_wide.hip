// gfx950 spread kernels for kernel widths 9..16 (tol < 1e-7: double precision in
// practice), 2-D and 3-D. Replaces, for those widths, the thread-per-point tile kernel
// (spread_tile_generic_kernel in nufft_kernels.hip; reference SpreadSubproblem2D/3DKernel,
// nufft_plan.cu.cc:790-960, 1295-1511, where one thread walks a point's whole stencil and the
// 64 unrelated points of a wavefront collide on LDS banks).
//
// One wavefront works on one point at a time. Lanes are 16 (x) by 4 (the slowest
// dimension: y in 2-D, z in 3-D); a ds_add_f64 wave-instruction therefore covers a
// 16 x 4 slab of the stencil, and the stencil takes ceil(W / 4) of them per component
// (2-D) or W ceil(W / 4) (3-D, looping over y). The stride of the lane-split dimension is
// 16 (mod 32) doubles -- the row stride 48 in 2-D, the padded plane stride in 3-D -- so the
// two 16-cell rows of a half-wave cover the 64 LDS banks exactly once: conflict free for
// every point position (the same rule as the 8 x 8 kernels of nufft_kernels.hip, whose
// stride is 8 mod 32 for four 8-cell rows).
//
// Accumulation is fp64 in LDS for both precisions (ds_add_f32 is 22x slower on this
// chip, EXPERIMENTS.md section 4). 3-D runs one launch per component (one fp64 plane of
// 45-85 KB in LDS), 2-D both in one launch (2 x 18 KB).
//
// Kernel values: CH points of a wave's share at a time, lane = (point, dimension)
// evaluates all 16 polynomial columns of one dimension of one point (wave-uniform
// coefficient loads); x and the lane-split dimension go through a small LDS staging
// area, the middle dimension of 3-D stays in the evaluating lane and is broadcast with
// v_readlane.
#include <cstdio>
#include <cstdlib>

#include "nufft_hip_internal.h"
#include "nufft_device.h"

namespace nufft_hip {

namespace {

template <int RANK, int W>
struct WideGeo {
  static constexpr int WR = (W + 3) / 4;   // 4-row groups of the lane-split dimension
  static constexpr int T0 = RANK == 2 ? 32 : (W <= 12 ? 16 : 8);
  static constexpr int T1 = RANK == 2 ? 32 : 8;
  static constexpr int T2 = RANK == 2 ? 1 : (W == 16 ? 2 : 4);   // (w = 16: the fp64 interp tile must fit 160 KB)
  static constexpr int L0 = T0 + W - 1, L1 = T1 + W - 1, L2 = RANK == 2 ? 1 : T2 + W - 1;
  static constexpr int LS = RANK == 2 ? 48 : L0;
  static constexpr int PS0 = LS * L1;
  static constexpr int PS = RANK == 2 ? PS0 : PS0 + (48 - PS0 % 32) % 32;   // 3-D: 16 (mod 32)
  static constexpr int SS = RANK == 2 ? LS : PS;                            // stride between the 4 lane rows
  static constexpr int plane = RANK == 2 ? PS0 : PS * L2;
  // lanes with dx >= W add 0 up to 15 cells past their row: behind the last row that is past the plane
  static constexpr int pad = 16;
};

constexpr int kWideNW = 8;
template <typename T, int RANK> constexpr int kWideCH = (sizeof(T) == 8 && RANK == 3) ? 8 : 16;   // staged points per wave
// staging row pitch (16 values + pad): consecutive lanes' 16-byte stores then start 36 / 20 banks
// apart and the 16 rows written by one ds_write_b128 cover the 64 banks once (pitch 16: 8- / 4-way conflicts)
template <typename T> constexpr int kWideRP = sizeof(T) == 8 ? 18 : 20;

template <typename T>
__device__ __forceinline__ void horner16(const T* __restrict__ tab, int nc, T z, T (&k)[16]) {
#pragma unroll
  for (int q = 0; q < 16; ++q) k[q] = tab[(nc - 1) * kMaxW + q];
  for (int t = nc - 2; t >= 0; --t) {
#pragma unroll
    for (int q = 0; q < 16; ++q) k[q] = fma_sgpr(k[q], z, tab[t * kMaxW + q]);
  }
}

// COMP: 0 = both components (two planes), 1 / 2 = real / imaginary part only (one plane).
// STACK (r06, 3-D): the workgroup walks a STACK of tiles consecutive in z (stack_plan_kernel, nufft_dense3.hip) instead
// of one subproblem: after every tile its T2 finished planes are written out, the W - 1 halo planes move down by T2
// (chains of ceil((T2 + W - 1) / T2) planes per (y, x) column) and the freed planes are zeroed. The tile + halo of a
// subproblem is 10.8 x the tile at W = 10 (25 x 17 x 13 cells for 16 x 8 x 4) and all of it goes to the fine grid as
// global fp64 atomics, per component: 256^3 modes, M = 1e7, tol 1e-9 spent 37 ms there (profiles/r06_c128_before.txt).
template <typename T, int RANK, int W, int COMP, bool STACK = false>
__global__ __launch_bounds__(kWideNW * 64) void spread_wide_kernel(
    Geom g, SortedPoints<T> sp, const T* __restrict__ horner, const T* __restrict__ c,
    T* __restrict__ fw, int64_t c_stride, int64_t fw_stride, T scale) {
  static_assert(!STACK || (RANK == 3 && COMP != 0), "stacks: the 3-D one-component launches");
  using G = WideGeo<RANK, W>;
  using T2 = typename Pair<T>::type;
  constexpr int NW = kWideNW, CH = kWideCH<T, RANK>, WR = G::WR, RP = kWideRP<T>;
  constexpr int LS = G::LS, PS = G::PS, SS = G::SS;
  constexpr int NPL = COMP == 0 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* plane_re = reinterpret_cast<double*>(smem_raw);
  double* plane_im = plane_re + (NPL == 2 ? G::plane : 0);
  T* stage_all = reinterpret_cast<T*>(plane_re + NPL * G::plane + G::pad);
  int tb = 0, p0 = 0, p1 = 0, slot = 0;
  StackDesc sd = {0, 0, 1, 0, 0};
  StackColumn scol = {0, 0, 0};
  if constexpr (STACK) {
    if ((int)blockIdx.x >= sp.seg_count[0]) return;
    sd = stack_load(sp.segs, blockIdx.x);
    scol = stack_column(g, sd.col);
    slot = scol.item * (int)gridDim.y + (int)blockIdx.y;
  } else {
    if (!locate_subproblem(g, sp.tile_start, sp.sub_start, blockIdx.x, &tb, &p0, &p1, &slot)) return;
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  for (int i = tid; i < NPL * G::plane + G::pad; i += NW * 64) plane_re[i] = 0.0;
  __syncthreads();

  const int nc = g.ncoef;
  T* kxs = stage_all + wave * (CH * 2 * RP);   // [CH][RP] x values
  T* kss = kxs + CH * RP;                      // [CH][RP] values of the lane-split dimension
  const int dx = lane & 15, r = lane >> 4;
  const bool in_x = dx < W;
  // Horner phase: lane = (point of the chunk, dimension)
  const int hq = lane & (CH - 1), hd = lane / CH;
  const T2* cc = reinterpret_cast<const T2*>(c) + (int64_t)slot * c_stride;
  T* out = fw + 2 * (int64_t)slot * fw_stride;
  for (int ti = 0; ti < (STACK ? sd.nz : 1); ++ti) {
  if constexpr (STACK) {
    p0 = sd.p0; p1 = sd.p1;
    if (p0 < 0) {
      const int t = stack_tile_index(g, scol, sd.z0 + ti);
      p0 = sp.tile_start[t];
      p1 = sp.tile_start[t + 1];
    }
  }
  const int npt = p1 - p0;
  const int share = (npt + NW - 1) / NW;
  const int wbeg = p0 + wave * share;
  const int wend = (wbeg + share < p1) ? wbeg + share : p1;

  for (int base = wbeg; base < wend; base += CH) {
    const int j = base + hq;
    int off = 0;
    T cre = (T)0, cim = (T)0;
    T kmid[W];   // 3-D: y values, live in the lanes with hd == 1
#pragma unroll
    for (int q = 0; q < W; ++q) kmid[q] = (T)0;
    if (hd < RANK) {
      T kv[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) kv[q] = (T)0;
      if (j < wend) {
        const PointView<T> rec = unpack_rec<T, RANK>(sp.rec[j]);
        const T z = hd == 0 ? rec.z0 : (hd == 1 ? rec.z1 : rec.z2);
        horner16<T>(horner, nc, z, kv);
        if (hd == 0) {
          const T2 cv = cc[rec.idx];
          cre = cv.x * scale;
          cim = cv.y * scale;
          off = (int)(rec.loc & 1023) + (int)((rec.loc >> 10) & 1023) * LS +
                (RANK > 2 ? (int)((rec.loc >> 20) & 1023) * PS : 0);
        }
      }
      if (hd == 0 || hd == RANK - 1) {
        T* dst = (hd == 0 ? kxs : kss) + hq * RP;
#pragma unroll
        for (int q = 0; q < 16; ++q) dst[q] = kv[q];
      }
      if (RANK > 2) {
#pragma unroll
        for (int q = 0; q < W; ++q) kmid[q] = kv[q];
      }
    }
    int npts = wend - base;
    if (npts > CH) npts = CH;
    // staged values of the NEXT point are requested before this point's atomics
    T a_n = kxs[dx];
    T ks_n[WR];
#pragma unroll
    for (int gq = 0; gq < WR; ++gq) ks_n[gq] = kss[(4 * gq + r) & 15];
    for (int q = 0; q < npts; ++q) {
      const T a = in_x ? a_n : (T)0;
      T ks[WR];
#pragma unroll
      for (int gq = 0; gq < WR; ++gq) ks[gq] = ks_n[gq];
      const int qn = (q + 1 < npts) ? q + 1 : q;
      a_n = kxs[qn * RP + dx];
#pragma unroll
      for (int gq = 0; gq < WR; ++gq) ks_n[gq] = kss[qn * RP + ((4 * gq + r) & 15)];
      const int o = __builtin_amdgcn_readlane(off, q) + r * SS + dx;
      const T c_re = bcast_lane(cre, q), c_im = bcast_lane(cim, q);
      double* pr = plane_re + o;
      double* pi = plane_im + o;
      if (RANK == 2) {
#pragma unroll
        for (int gq = 0; gq < WR; ++gq) {
          const T v = a * ks[gq];
          // the last group is partial unless W is a multiple of 4: rows past the stencil are masked
          if (4 * gq + 3 < W || 4 * gq + r < W) {
            if (COMP != 2) lds_add(pr + 4 * gq * SS, (double)(v * c_re));
            if (COMP != 1) lds_add(pi + 4 * gq * SS, (double)(v * c_im));
          }
        }
      } else {
        T are[WR], aim[WR];
#pragma unroll
        for (int gq = 0; gq < WR; ++gq) {
          const T v = a * ks[gq];
          are[gq] = v * c_re;
          aim[gq] = v * c_im;
        }
#pragma unroll
        for (int dy = 0; dy < W; ++dy) {
          const T kyq = bcast_lane(kmid[dy], CH + q);
#pragma unroll
          for (int gq = 0; gq < WR; ++gq) {
            if (4 * gq + 3 < W || 4 * gq + r < W) {
              if (COMP != 2) lds_add(pr + dy * LS + 4 * gq * SS, (double)(are[gq] * kyq));
              if (COMP != 1) lds_add(pi + dy * LS + 4 * gq * SS, (double)(aim[gq] * kyq));
            }
          }
        }
      }
    }
  }
  __syncthreads();
  if constexpr (STACK) {
    // tile sd.z0 + ti is complete in its first T2 planes (the last tile of the stack: in all of them)
    const bool last = ti == sd.nz - 1;
    const int o0 = scol.t0 * G::T0, o1 = scol.t1 * G::T1, o2 = (sd.z0 + ti) * G::T2;
    const int nrows = (last ? G::L2 : G::T2) * G::L1;
    const int a0 = lane < G::L0 ? lane : G::L0 - 1;
    const bool lane_on = lane < G::L0;
    const int gx = wrap1(o0 + a0, g.nf[0]);
    for (int rho = wave; rho < nrows; rho += NW) {
      const int a2 = rho / G::L1, a1 = rho - a2 * G::L1;
      const int lrow = a2 * PS + a1 * LS + a0;
      const int64_t gbase = (int64_t)g.nf[0] * (wrap1(o1 + a1, g.nf[1]) + (int64_t)g.nf[1] * wrap1(o2 + a2, g.nf[2]));
      constexpr int CHAIN = (G::L2 + G::T2 - 1) / G::T2;   // planes a2, a2 + T2, ... of this (y, x) column
      double v[CHAIN];
#pragma unroll
      for (int m = 0; m < CHAIN; ++m) v[m] = (m == 0 || (!last && a2 + m * G::T2 < G::L2)) ? plane_re[lrow + m * G::T2 * PS] : 0.0;
      if (lane_on && v[0] != 0.0) glb_add(&out[2 * (gbase + gx) + (COMP - 1)], (T)v[0]);
      if (!last && lane_on) {
#pragma unroll
        for (int m = 0; m < CHAIN; ++m)
          if (a2 + m * G::T2 < G::L2) plane_re[lrow + m * G::T2 * PS] = m + 1 < CHAIN ? v[m + 1] : 0.0;
      }
    }
    if (!last) __syncthreads();
    continue;
  }

  // write-out: add the tile to the periodic fine grid
  int t0, t1, t2;
  tile_coords(g, tb, &t0, &t1, &t2);
  const int o0 = t0 * G::T0, o1 = t1 * G::T1, o2 = t2 * G::T2;
  for (RowWalk rw(wave, G::L1); rw.a2 < G::L2; rw.advance(NW, G::L1)) {
    const int g1 = wrap1(o1 + rw.a1, g.nf[1]);
    const int g2 = RANK > 2 ? wrap1(o2 + rw.a2, g.nf[2]) : 0;
    const int64_t rowbase = (int64_t)g.nf[0] * (g1 + (int64_t)g.nf[1] * g2);
    const int lrow = rw.a2 * PS + rw.a1 * LS;
    if (COMP == 0) {
      for (int e = lane; e < 2 * G::L0; e += 64) {
        const int a0 = e >> 1, comp = e & 1;
        const T v = (T)(comp ? plane_im : plane_re)[lrow + a0];
        if (v != (T)0) glb_add(&out[2 * (rowbase + wrap1(o0 + a0, g.nf[0])) + comp], v);
      }
    } else {
      if (lane < G::L0) {
        const T v = (T)plane_re[lrow + lane];
        if (v != (T)0) glb_add(&out[2 * (rowbase + wrap1(o0 + lane, g.nf[0])) + (COMP - 1)], v);
      }
    }
  }
  }   // (tiles of the stack; one pass otherwise)
}

// ------------------------------------------------------------ interp, widths 9..16
//
// Replaces, for these widths, the gather-from-global kernel (interp_tile_generic_kernel;
// reference InterpSubproblem2D/3DKernel, nufft_plan.cu.cc:1041-1187, 1608-1804). The tile
// (with its halo) is loaded into LDS as interleaved complex T; lanes are 16 (x) by 4 POINTS:
// a ds_read_b128 / b64 wave-instruction fetches one 16-cell stencil row of four points, each
// quarter-wave reading 256 (128) contiguous bytes -- conflict free whatever the four positions,
// since a wide LDS read is served one quarter-wave at a time. Each lane keeps its point's y
// (and z) kernel values in registers, multiplies and accumulates; the 16 lanes of a point are
// summed with four row_shr DPP steps. 2-D: 16 points (4 quads) are staged per wave at a time,
// 3-D: 4 (LDS is the tile's: 100-158 KB in fp64).
template <int RANK> constexpr int kWideIQ = RANK == 2 ? 4 : 1;

template <int CTRL>
__device__ __forceinline__ float dpp_mov0(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ double dpp_mov0(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
// lane 15 of every row of 16 lanes ends up with the row's sum (row_shr 1, 2, 4, 8; lanes shifted in are 0)
template <typename T>
__device__ __forceinline__ T row_sum16(T v) {
  v += dpp_mov0<0x111>(v);
  v += dpp_mov0<0x112>(v);
  v += dpp_mov0<0x114>(v);
  v += dpp_mov0<0x118>(v);
  return v;
}

// STACK (r06, 3-D): the workgroup walks a STACK of tiles consecutive in z (stack_plan_kernel, nufft_dense3.hip): the planes a
// tile shares with the next one move down in LDS, the next tile's T2 new planes are requested into registers before this
// tile's points (the pipelined form of interp_point_kernel<..., STACK>, nufft_kernels.hip). Tile + halo is 10.8 x the tile
// at W = 10 (25 x 17 x 13 cells for 16 x 8 x 4), read per subproblem: 256^3 modes, tol 1e-9, M = 1e7: 15.3 ms.
template <typename T, int RANK, int W, bool STACK = false>
__global__ __launch_bounds__(kWideNW * 64) void interp_wide_kernel(
    Geom g, SortedPoints<T> sp, const T* __restrict__ horner, T* __restrict__ c,
    const T* __restrict__ fw, int64_t c_stride, int64_t fw_stride, T scale) {
  static_assert(!STACK || RANK == 3, "stacks: 3-D");
  using G = WideGeo<RANK, W>;
  using T2 = typename Pair<T>::type;
  constexpr int NW = kWideNW, NQ = kWideIQ<RANK>, CH = 4 * NQ, RP = kWideRP<T>;
  constexpr int L0 = G::L0, L1 = G::L1, L2 = G::L2;
  constexpr int LS = L0, PS = LS * L1, cells = PS * L2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T2* tile = reinterpret_cast<T2*>(smem_raw);
  T* stage_all = reinterpret_cast<T*>(tile + cells + 16);
  int tb = 0, p0 = 0, p1 = 0, slot = 0;
  int t0 = 0, t1 = 0, t2 = 0;
  StackDesc sd = {0, 0, 1, 0, 0};
  StackColumn scol = {0, 0, 0};
  if constexpr (STACK) {
    if ((int)blockIdx.x >= sp.seg_count[0]) return;
    sd = stack_load(sp.segs, blockIdx.x);
    scol = stack_column(g, sd.col);
    slot = scol.item * (int)gridDim.y + (int)blockIdx.y;
    t0 = scol.t0; t1 = scol.t1; t2 = sd.z0;
  } else {
    if (!locate_subproblem(g, sp.tile_start, sp.sub_start, blockIdx.x, &tb, &p0, &p1, &slot)) return;
    tile_coords(g, tb, &t0, &t1, &t2);
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int o0 = t0 * G::T0, o1 = t1 * G::T1, o2 = t2 * G::T2;
  const T2* in = reinterpret_cast<const T2*>(fw) + (int64_t)slot * fw_stride;
  // tile + halo -> LDS: cells are dealt to threads in flat order (rows are only 24-47 cells long: a
  // lane-per-column walk would leave half of every wavefront idle); kBatch loads in flight before
  // the stores, so a workgroup pays two or three memory round trips for its 35-160 KB tile
  {
    constexpr int kBatch = 8;
    for (int e0 = tid; e0 < cells; e0 += kBatch * NW * 64) {
      T2 v[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int e = e0 + u * NW * 64;
        const int ec = e < cells ? e : cells - 1;
        const int rowi = ec / L0;          // (compile-time divisors)
        const int a0 = ec - rowi * L0;
        const int a2 = RANK > 2 ? rowi / L1 : 0;
        const int a1 = rowi - a2 * L1;
        const int64_t gx = wrap1(o0 + a0, g.nf[0]);
        const int g1 = wrap1(o1 + a1, g.nf[1]);
        const int g2 = RANK > 2 ? wrap1(o2 + a2, g.nf[2]) : 0;
        v[u] = in[(int64_t)g.nf[0] * (g1 + (int64_t)g.nf[1] * g2) + gx];
      }
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        const int e = e0 + u * NW * 64;
        if (e < cells) tile[e] = v[u];
      }
    }
    // lanes with dx >= W read (with weight 0) up to 15 cells past their row: keep those finite
    if (tid < 16) { T2 z; z.x = (T)0; z.y = (T)0; tile[cells + tid] = z; }
  }
  __syncthreads();

  const int nc = g.ncoef;
  T* kst = stage_all + wave * (RANK * CH * RP);   // [RANK][CH][RP]
  const int dx = lane & 15, pr = lane >> 4;
  const bool in_x = dx < W;
  const int hq = lane % CH, hd = lane / CH;       // Horner phase: lane = (point of the chunk, dimension)
  T2* cc = reinterpret_cast<T2*>(c) + (int64_t)slot * c_stride;
  auto do_points = [&](int p0, int p1) {
  const int npt = p1 - p0;
    const int share = (npt + NW - 1) / NW;
    const int wbeg = p0 + wave * share;
    const int wend = (wbeg + share < p1) ? wbeg + share : p1;

    for (int base = wbeg; base < wend; base += CH) {
      int off = 0, idx = 0;
      if (hd < RANK) {
        T kv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) kv[q] = (T)0;
        const int j = base + hq;
        if (j < wend) {
          const PointView<T> rec = unpack_rec<T, RANK>(sp.rec[j]);
          const T z = hd == 0 ? rec.z0 : (hd == 1 ? rec.z1 : rec.z2);
          horner16<T>(horner, nc, z, kv);
          idx = rec.idx;
          off = (int)(rec.loc & 1023) + (int)((rec.loc >> 10) & 1023) * LS +
                (RANK > 2 ? (int)((rec.loc >> 20) & 1023) * PS : 0);
        }
        T* dst = kst + (hd * CH + hq) * RP;
#pragma unroll
        for (int q = 0; q < 16; ++q) dst[q] = kv[q];
      }
#pragma unroll 1
      for (int qq = 0; qq < NQ; ++qq) {
        const int p = 4 * qq + pr;   // this lane row's point of the chunk (lane p, dimension 0, holds off / idx)
        if (base + 4 * qq >= wend) break;
        const int offp = __shfl(off, p);
        const int idxp = __shfl(idx, p);
        const T a0 = kst[p * RP + dx];
        const T a = in_x ? a0 : (T)0;
        T ky[W];
#pragma unroll
        for (int q = 0; q < W; ++q) ky[q] = kst[(CH + p) * RP + q];
        const T2* tp = tile + offp + dx;
        T sre = (T)0, sim = (T)0;
        if (RANK == 2) {
#pragma unroll
          for (int dy = 0; dy < W; ++dy) {
            const T2 v = lds_cell(tp + dy * LS);
            const T wgt = a * ky[dy];
            sre = fma(wgt, v.x, sre);
            sim = fma(wgt, v.y, sim);
          }
        } else {
          T kz[W];
#pragma unroll
          for (int q = 0; q < W; ++q) kz[q] = kst[(2 * CH + p) * RP + q];
#pragma unroll
          for (int dz = 0; dz < W; ++dz) {
            const T az = a * kz[dz];
#pragma unroll
            for (int dy = 0; dy < W; ++dy) {
              const T2 v = lds_cell(tp + dz * PS + dy * LS);
              const T wgt = az * ky[dy];
              sre = fma(wgt, v.x, sre);
              sim = fma(wgt, v.y, sim);
            }
          }
        }
        sre = row_sum16(sre);
        sim = row_sum16(sim);
        if (dx == 15 && base + p < wend) {
          T2 out;
          out.x = sre * scale;
          out.y = sim * scale;
          cc[idxp] = out;
        }
      }
    }
  };
  if constexpr (STACK) {
    constexpr int NT = NW * 64;
    constexpr int NEWC = PS * G::T2;                         // cells of a tile's new planes
    constexpr int NPF = (NEWC + NT - 1) / NT;
    static_assert(NPF <= 8, "prefetch registers");
    constexpr int kRng = 64;
    __shared__ int rng[2 * kRng];
    if (sd.p0 < 0) {
      for (int i = tid; i < sd.nz && i < kRng; i += NT) {
        const int t = stack_tile_index(g, scol, sd.z0 + i);
        rng[2 * i] = sp.tile_start[t];
        rng[2 * i + 1] = sp.tile_start[t + 1];
      }
    }
    __syncthreads();
    auto range_of = [&](int i, int* q0, int* q1) {
      if (sd.p0 >= 0) { *q0 = sd.p0; *q1 = sd.p1; return; }   // a piece: one tile, its own points
      if (i < kRng) { *q0 = rng[2 * i]; *q1 = rng[2 * i + 1]; return; }
      const int t = stack_tile_index(g, scol, sd.z0 + i);
      *q0 = sp.tile_start[t];
      *q1 = sp.tile_start[t + 1];
    };
    for (int ti = 0; ti < sd.nz; ++ti) {
      const bool more = ti + 1 < sd.nz;
      int c0, c1;
      range_of(ti, &c0, &c1);
      // the next tile's new planes in named registers (see interp_point_kernel); behind the last tile the planes it already
      // holds are read again and dropped
      T2 pf0, pf1, pf2, pf3, pf4, pf5, pf6, pf7;
      const int o2n = (sd.z0 + (more ? ti + 1 : ti)) * G::T2 + (L2 - G::T2);
      auto pf_src = [&](int u) {
        const int e = tid + u * NT;
        const int ec = e < NEWC ? e : NEWC - 1;
        const int rowi = ec / L0, a0 = ec - rowi * L0;
        const int a2 = rowi / L1, a1 = rowi - a2 * L1;
        return in[(int64_t)g.nf[0] * (wrap1(o1 + a1, g.nf[1]) + (int64_t)g.nf[1] * wrap1(o2n + a2, g.nf[2])) + wrap1(o0 + a0, g.nf[0])];
      };
#define NUFFT_PF_LOAD(u) if constexpr (u < NPF) pf##u = pf_src(u); else pf##u = T2();
      NUFFT_PF_LOAD(0) NUFFT_PF_LOAD(1) NUFFT_PF_LOAD(2) NUFFT_PF_LOAD(3) NUFFT_PF_LOAD(4) NUFFT_PF_LOAD(5) NUFFT_PF_LOAD(6) NUFFT_PF_LOAD(7)
#undef NUFFT_PF_LOAD
      do_points(c0, c1);
      if (!more) break;
      __syncthreads();   // every thread is done with this tile's planes
      // planes T2 .. L2 - 1 move down by T2: one thread per (y, x) cell and residue of the plane index, upwards
      for (int e = tid; e < NEWC; e += NT) {
        const int a2 = e / PS, r = e - a2 * PS;
        T2* col = tile + r;
        for (int q = a2; q + G::T2 < L2; q += G::T2) col[q * PS] = col[(q + G::T2) * PS];
      }
      __syncthreads();   // (the new planes land where the moved ones were read)
      auto pf_dst = [&](int u, const T2& v) {
        const int e = tid + u * NT;
        if (e < NEWC) tile[(L2 - G::T2) * PS + e] = v;
      };
#define NUFFT_PF_STORE(u) if constexpr (u < NPF) pf_dst(u, pf##u);
      NUFFT_PF_STORE(0) NUFFT_PF_STORE(1) NUFFT_PF_STORE(2) NUFFT_PF_STORE(3) NUFFT_PF_STORE(4) NUFFT_PF_STORE(5) NUFFT_PF_STORE(6) NUFFT_PF_STORE(7)
#undef NUFFT_PF_STORE
      __syncthreads();
    }
    return;
  }
  do_points(p0, p1);
}

template <typename K>
hipError_t wide_ensure_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             (int)bytes);
}

template <typename T, int RANK, int W, int COMP>
hipError_t launch_one(const Geom& g, dim3 grid, const SortedPoints<T>& sp, const T* horner, const T* c, T* fw,
                      int64_t c_stride, int64_t fw_stride, T scale, hipStream_t stream) {
  using G = WideGeo<RANK, W>;
  constexpr size_t lds = sizeof(double) * ((COMP == 0 ? 2 : 1) * G::plane + G::pad) +
                         sizeof(T) * kWideNW * kWideCH<T, RANK> * 2 * kWideRP<T>;
  hipError_t e = wide_ensure_lds(spread_wide_kernel<T, RANK, W, COMP>, lds);
  if (e != hipSuccess) return e;
  spread_wide_kernel<T, RANK, W, COMP><<<grid, kWideNW * 64, lds, stream>>>(g, sp, horner, c, fw, c_stride,
                                                                            fw_stride, scale);
  return hipGetLastError();
}

template <typename T, int W, int COMP>
hipError_t launch_one_stack(const Geom& g, dim3 grid, const SortedPoints<T>& sp, const T* horner, const T* c, T* fw,
                            int64_t c_stride, int64_t fw_stride, T scale, hipStream_t stream) {
  using G = WideGeo<3, W>;
  constexpr size_t lds = sizeof(double) * (G::plane + G::pad) + sizeof(T) * kWideNW * kWideCH<T, 3> * 2 * kWideRP<T>;
  hipError_t e = wide_ensure_lds(spread_wide_kernel<T, 3, W, COMP, true>, lds);
  if (e != hipSuccess) return e;
  spread_wide_kernel<T, 3, W, COMP, true><<<grid, kWideNW * 64, lds, stream>>>(g, sp, horner, c, fw, c_stride, fw_stride, scale);
  return hipGetLastError();
}

template <typename T, int W>
hipError_t launch_w(const Geom& g, dim3 grid, const SortedPoints<T>& sp, const T* horner, const T* c, T* fw,
                    int64_t c_stride, int64_t fw_stride, T scale, hipStream_t stream, int64_t M) {
  if (g.rank == 2) return launch_one<T, 2, W, 0>(g, grid, sp, horner, c, fw, c_stride, fw_stride, scale, stream);
  if (g.stack && sp.segs) {   // r06: stacks of tiles
    const dim3 sgrid(stack_grid_bound(g, M), grid.y);
    hipError_t e = launch_one_stack<T, W, 1>(g, sgrid, sp, horner, c, fw, c_stride, fw_stride, scale, stream);
    if (e != hipSuccess) return e;
    return launch_one_stack<T, W, 2>(g, sgrid, sp, horner, c, fw, c_stride, fw_stride, scale, stream);
  }
  hipError_t e = launch_one<T, 3, W, 1>(g, grid, sp, horner, c, fw, c_stride, fw_stride, scale, stream);
  if (e != hipSuccess) return e;
  return launch_one<T, 3, W, 2>(g, grid, sp, horner, c, fw, c_stride, fw_stride, scale, stream);
}

}  // namespace

// Host-side mirror of WideGeo: the tile the plan must sort by for these kernels.
bool wide_spread_supported(int rank, int w) { return (rank == 2 || rank == 3) && w > 8 && w <= 16; }
void wide_spread_tile(int rank, int w, int tile[3]) {
  tile[0] = rank == 2 ? 32 : (w <= 12 ? 16 : 8);
  tile[1] = rank == 2 ? 32 : 8;
  tile[2] = rank == 2 ? 1 : (w == 16 ? 2 : 4);
}
int wide_spread_lstride(int rank, int w) { return rank == 2 ? 48 : (w <= 12 ? 16 : 8) + w - 1; }
size_t wide_spread_lds_bytes(int rank, int w, int precision) {
  int t[3];
  wide_spread_tile(rank, w, t);
  const int ls = wide_spread_lstride(rank, w), l1 = t[1] + w - 1;
  size_t plane = (size_t)ls * l1;
  if (rank == 3) {
    plane += (48 - plane % 32) % 32;
    plane *= (size_t)(t[2] + w - 1);
  }
  return sizeof(double) * ((rank == 2 ? 2 : 1) * plane + 16) + (size_t)precision * kWideNW * (precision == 8 ? (rank == 3 ? 8 : 16) * 2 * 18 : 16 * 2 * 20);
}

size_t wide_interp_lds_bytes(int rank, int w, int precision) {
  int t[3];
  wide_spread_tile(rank, w, t);
  size_t cells = 1;
  for (int d = 0; d < rank; ++d) cells *= (size_t)(t[d] + w - 1);
  return (cells + 16) * 2 * (size_t)precision +
         (size_t)precision * kWideNW * rank * 4 * (rank == 2 ? 4 : 1) * (precision == 8 ? 18 : 20);
}

template <typename T>
hipError_t launch_interp_wide(const Geom& g, const SortedPoints<T>& sp, int64_t M, const T* horner, T* c,
                              const T* fw, int batch, int64_t c_stride, int64_t fw_stride, T scale,
                              hipStream_t stream) {
  if (M == 0) return hipSuccess;
  dim3 grid((unsigned)((int64_t)g.ntiles + M / g.max_sub), (unsigned)batch);
  const size_t lds = wide_interp_lds_bytes(g.rank, g.w, (int)sizeof(T));
  hipError_t e = hipSuccess;
  if (g.rank == 3 && g.stack && sp.segs) {   // r06: over stacks of tiles
    const dim3 sgrid(stack_grid_bound(g, M), (unsigned)batch);
#define NUFFT_WIDE_IPS(WW)                                                                           \
  case WW:                                                                                           \
    e = wide_ensure_lds(interp_wide_kernel<T, 3, WW, true>, lds);                                    \
    if (e != hipSuccess) return e;                                                                   \
    interp_wide_kernel<T, 3, WW, true><<<sgrid, kWideNW * 64, lds, stream>>>(g, sp, horner, c, fw, c_stride, fw_stride, scale); \
    break;
    switch (g.w) {
      NUFFT_WIDE_IPS(9) NUFFT_WIDE_IPS(10) NUFFT_WIDE_IPS(11) NUFFT_WIDE_IPS(12)
      NUFFT_WIDE_IPS(13) NUFFT_WIDE_IPS(14) NUFFT_WIDE_IPS(15) NUFFT_WIDE_IPS(16)
      default: return hipErrorInvalidValue;
    }
#undef NUFFT_WIDE_IPS
    return hipGetLastError();
  }
#define NUFFT_WIDE_IP(RR, WW)                                                                        \
  case RR * 100 + WW:                                                                                \
    e = wide_ensure_lds(interp_wide_kernel<T, RR, WW>, lds);                                         \
    if (e != hipSuccess) return e;                                                                   \
    interp_wide_kernel<T, RR, WW><<<grid, kWideNW * 64, lds, stream>>>(g, sp, horner, c, fw, c_stride, \
                                                                       fw_stride, scale);            \
    break;
  switch (g.rank * 100 + g.w) {
    NUFFT_WIDE_IP(2, 9) NUFFT_WIDE_IP(2, 10) NUFFT_WIDE_IP(2, 11) NUFFT_WIDE_IP(2, 12)
    NUFFT_WIDE_IP(2, 13) NUFFT_WIDE_IP(2, 14) NUFFT_WIDE_IP(2, 15) NUFFT_WIDE_IP(2, 16)
    NUFFT_WIDE_IP(3, 9) NUFFT_WIDE_IP(3, 10) NUFFT_WIDE_IP(3, 11) NUFFT_WIDE_IP(3, 12)
    NUFFT_WIDE_IP(3, 13) NUFFT_WIDE_IP(3, 14) NUFFT_WIDE_IP(3, 15) NUFFT_WIDE_IP(3, 16)
    default: return hipErrorInvalidValue;
  }
#undef NUFFT_WIDE_IP
  return hipGetLastError();
}
template hipError_t launch_interp_wide<float>(const Geom&, const SortedPoints<float>&, int64_t, const float*,
                                              float*, const float*, int, int64_t, int64_t, float, hipStream_t);
template hipError_t launch_interp_wide<double>(const Geom&, const SortedPoints<double>&, int64_t, const double*,
                                               double*, const double*, int, int64_t, int64_t, double,
                                               hipStream_t);

template <typename T>
hipError_t launch_spread_wide(const Geom& g, const SortedPoints<T>& sp, int64_t M, const T* horner, const T* c,
                              T* fw, int batch, int64_t c_stride, int64_t fw_stride, T scale,
                              hipStream_t stream) {
  if (M == 0) return hipSuccess;
  dim3 grid((unsigned)((int64_t)g.ntiles + M / g.max_sub), (unsigned)batch);
#define NUFFT_WIDE_CASE(WW) \
  case WW: return launch_w<T, WW>(g, grid, sp, horner, c, fw, c_stride, fw_stride, scale, stream, M);
  switch (g.w) {
    NUFFT_WIDE_CASE(9) NUFFT_WIDE_CASE(10) NUFFT_WIDE_CASE(11) NUFFT_WIDE_CASE(12)
    NUFFT_WIDE_CASE(13) NUFFT_WIDE_CASE(14) NUFFT_WIDE_CASE(15) NUFFT_WIDE_CASE(16)
    default: return hipErrorInvalidValue;
  }
#undef NUFFT_WIDE_CASE
}
template hipError_t launch_spread_wide<float>(const Geom&, const SortedPoints<float>&, int64_t, const float*,
                                              const float*, float*, int, int64_t, int64_t, float, hipStream_t);
template hipError_t launch_spread_wide<double>(const Geom&, const SortedPoints<double>&, int64_t, const double*,
                                               const double*, double*, int, int64_t, int64_t, double,
                                               hipStream_t);

}  // namespace nufft_hip
