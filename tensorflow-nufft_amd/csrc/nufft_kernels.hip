// gfx950 (MI355X, CDNA4) kernels of the NUFFT plan: point preparation, tile
// sort, exponential-of-semicircle spreading (type 1) and interpolation
// (type 2), and the fused deconvolution / mode-reordering step.
//
// Replaces the CUDA kernels of the reference's
// tensorflow_nufft/cc/kernels/nufft_plan.cu.cc (inventory: SURVEY.md 2.1):
//   CalcBinSizeNoGhost*/CalcInvertofGlobalSortIdx*  (:160-296)  -> prep / scan / scatter
//   CalcSubproblem / MapBinToSubproblem            (:304-320)  -> scan + in-kernel search
//   SpreadSubproblem*                               (:530-960, :1295-1511) -> spread_tile_*
//   InterpNuptsDriven* / InterpSubproblem*          (:653-704, :963-1187, :1513-1804) -> interp_tile_*
//   Deconvolve* / Amplify*                          (:326-435)  -> deconvolve_kernel
// None of it is a translation: the data layout (tile-local stencil starts +
// fractional Horner arguments computed in double), the binning rule (by
// stencil start, so the halo is one sided), the subproblem lookup (binary
// search in a scanned array, no host sync) and the wavefront-per-point
// scatter are specific to this build. Wavefront = 64 lanes throughout.
//
// Map of this file (EXPERIMENTS.md sections 3-5 have the measurements). Kernels for widths 9-16 live in
// nufft_wide.hip, the 1-D interpolation in nufft_line.hip, the pruned FFT passes in nufft_fft.hip; the
// device helpers they share (record decoding, RowWalk, locate_subproblem, LDS / global adds) in
// nufft_device.h.
//   helpers            horner8 / hornerW, record packing, tile_to_grid
//   sort               fold_coords; hist_lds / colscan / scan_tiles / scatter_lds (<= 16384 tiles),
//                      hist16_lds / colscan16 / scatter_ranked (<= 73728 tiles), count_global /
//                      scatter_global beyond; cellsort2d / cellsort3d (second level, by start cell)
//   spread (type 1)    spread_2d_w8_group_kernel (2-D, w <= 8, dense: the config-2 kernel),
//                      spread_2d_w8_wave_kernel (2-D float w = 8, sparse), spread_wave2_kernel
//                      (2-D, sparse, other widths / double), spread_wave3_kernel (3-D, fp64 planes
//                      or packed fixed point; crowded tiles of fixed-point plans fall back to the fp64
//                      planes), spread_tile_generic_kernel (1-D, explicit TILE_GENERIC, tiny grids)
//   interp (type 2)    interp_point_kernel (LDS tile, thread per point), interp_tile_generic_kernel
//   deconvolve_kernel, permute_kernel; launchers and the launch-shape / LDS-size rules at the end
#include <algorithm>
#include <mutex>
#include <type_traits>
#include <cstdio>
#include <cstdlib>

#include "nufft_hip_internal.h"
#include "nufft_device.h"

namespace nufft_hip {

namespace {

constexpr double kPiD = 3.14159265358979323846;

// ----------------------------------------------------------------- helpers

template <typename T>
__device__ __forceinline__ T horner_cell(const T* __restrict__ tab, int nc, int j, T z) {
  T acc = tab[(nc - 1) * kMaxW + j];
  for (int k = nc - 2; k >= 0; --k) acc = fma(acc, z, tab[k * kMaxW + j]);
  return acc;
}

// Piecewise-polynomial kernel values for the first 8 stencil cells of up to
// three dimensions at once. The coefficient loop is OUTER, so each step is one
// wave-uniform 8-wide row load (s_load_dwordx8) feeding 8 x NDIM independent
// FMAs; with the cell loop outer the compiler emits one scalar load per FMA and
// waits for each (measured: 150 us of a 365 us interp kernel). Rows are zero
// beyond the kernel width, so cells q >= w come out exactly 0.
template <typename T, int NDIM>
__device__ __forceinline__ void horner8_rt(const T* __restrict__ tab, int nc, T z0, T z1, T z2,
                                           T (&k0)[8], T (&k1)[8], T (&k2)[8]) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const T t = tab[(nc - 1) * kMaxW + q];
    k0[q] = t; k1[q] = t; k2[q] = t;
  }
  for (int k = nc - 2; k >= 0; --k) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const T t = tab[k * kMaxW + q];
      k0[q] = fma(k0[q], z0, t);
      if (NDIM > 1) k1[q] = fma(k1[q], z1, t);
      if (NDIM > 2) k2[q] = fma(k2[q], z2, t);
    }
  }
}

// Every width <= 8 needs at most kFixedCoef terms in either precision (rows
// above the fitted count are zero), so the common case is a fully unrolled
// evaluation whose row loads have no loop-carried wait: they are issued together
// (or hoisted out of the point loop when the scalar registers allow).
constexpr int kFixedCoef = 10;
template <typename T, int NDIM>
__device__ __forceinline__ void horner8(const T* __restrict__ tab, int nc, T z0, T z1, T z2,
                                        T (&k0)[8], T (&k1)[8], T (&k2)[8]) {
  if (nc > kFixedCoef) {
    horner8_rt<T, NDIM>(tab, nc, z0, z1, z2, k0, k1, k2);
    return;
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const T t = tab[(kFixedCoef - 1) * kMaxW + q];
    k0[q] = t; k1[q] = t; k2[q] = t;
  }
#pragma unroll
  for (int k = kFixedCoef - 2; k >= 0; --k) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const T t = tab[k * kMaxW + q];
      k0[q] = fma(k0[q], z0, t);
      if (NDIM > 1) k1[q] = fma(k1[q], z1, t);
      if (NDIM > 2) k2[q] = fma(k2[q], z2, t);
    }
  }
}

// ------------------------------------------------------------ prep + sort

// Fold + rescale one point in DOUBLE (reference FoldAndRescale,
// nufft_plan.h:676-734, does it in FloatType, which costs ~eps*nf cells of
// position error in float), split into the integer stencil start
// i0 = ceil(x' - w/2) (reference nufft_plan.cu.cc:838-841 / nufft_plan.cc:1496)
// and the Horner argument z = 2(i0 - x') + w - 1 in [-1, 1]. Points are binned
// by the WRAPPED STENCIL START, so a tile's halo is one sided (w - 1 cells).
// Returns the tile index; fills the record (without idx).
// Coordinates of point i. All three loads are unconditional (the host points unused
// dimensions at the x array): loads under a branch are waited for one by one, which
// defeats the batching in the count / scatter loops below.
// AOS = 2 / 3: the coordinates are the columns of one [M, rank] array with x LAST (what the
// op hands over, nufft_kernels.cc:282-286 reversed here by the pointer order): one 8- or
// 12-byte load per point instead of two or three strided 4-byte ones.
template <typename T, int AOS>
__device__ __forceinline__ void load_coords(const PointsIn& in, int64_t i, T x[3]) {
  if constexpr (AOS == 2) {
    typedef T v2 __attribute__((ext_vector_type(2)));
    const v2 v = reinterpret_cast<const v2*>(in.pts[1])[i];
    x[0] = v.y; x[1] = v.x; x[2] = v.y;
  } else if constexpr (AOS == 3) {
    const T* b = reinterpret_cast<const T*>(in.pts[2]) + 3 * i;
    x[2] = b[0]; x[1] = b[1]; x[0] = b[2];
  } else {
#pragma unroll
    for (int d = 0; d < 3; ++d) x[d] = ((const T*)in.pts[d])[i * in.stride];
  }
}

// QF (quick fold): the caller has checked quick_fold(g, in) on the host.
// double -> int with the hardware's semantics: saturating, NaN -> 0 (defined for every input, unlike the C++ cast)
__device__ __forceinline__ int cvt_i32_sat(double x) {
  int r;
  asm("v_cvt_i32_f64 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
template <typename T, bool QF = false>
__device__ __forceinline__ int fold_coords(const Geom& g, const PointsIn& in, const T xin[3], Rec<T>* r,
                                           bool* bad) {
  uint32_t loc = 0;
  int tc[3] = {0, 0, 0};
  T zz3[3] = {(T)0, (T)0, (T)0};
  // Common case, compiled separately: the points are promised inside [-pi, pi] or [-3 pi, 3 pi] and every tile edge is
  // a power of two -- no fmod, no division, no 64-bit modulo; the stencil start of an in-range point needs at most one
  // wrap. Garbage (out-of-range points with the check switched off, NaN, Inf) stays memory-safe through the
  // saturating conversion and the clamp. Same arithmetic as the general path below for every valid point.
  // r04 A/B (config 2 / config 4): count 36.5 -> 26 us / 393 -> 232 us, level-1 scatter 1.96 -> 1.82 ms; as a
  // workgroup-uniform run-time branch in ONE kernel body it gave 32.7 / 341 us and nothing in the scatter.
  if constexpr (QF) {
    // (straight-line: the EXTENDED fold is the STRICT one for points inside [-pi, pi]; only the limit of the range
    // check depends on the mode)
    const double lim = in.range_mode == NUFFT_HIP_RANGE_STRICT ? kPiD : 3.0 * kPiD;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (d >= g.rank) break;
      const double x = (double)xin[d];
      *bad |= !(x > -lim && x < lim);
      const double s = (x > kPiD) ? x - kPiD : ((x < -kPiD) ? x + 3.0 * kPiD : x + kPiD);
      const int nf = g.nf[d];
      const double xs = s * ((double)nf * (1.0 / (2.0 * kPiD)));
      const double i0f = ceil(xs - 0.5 * (double)g.w);   // in [-w / 2, nf) for a point in range
      double zz = 2.0 * (i0f - xs) + (double)(g.w - 1);
      zz = fmin(1.0, fmax(-1.0, zz));
      // garbage coordinates (NaN, Inf, 1e30 with the range check off): the conversion is the INSTRUCTION, which
      // saturates and maps NaN to 0, not the C++ cast, whose result is undefined out of range -- the memory safety of
      // the scattered stores must not rest on how today's compiler lowers a cast (r04 advisor; a clamp in double cost
      // the two hottest scatter instantiations 20-28 bytes of scratch)
      int i0 = cvt_i32_sat(i0f);
      if (i0 < 0) i0 += nf;
      i0 = i0 < 0 ? 0 : (i0 >= nf ? nf - 1 : i0);
      tc[d] = i0 >> g.tile_shift[d];
      loc |= (uint32_t)(i0 & (g.tile[d] - 1)) << (10 * d);
      zz3[d] = (T)zz;
    }
    r->loc = loc;
    r->z0 = zz3[0];
    r->z1 = zz3[1];
    r->z2 = zz3[2];
    return tile_id(g, tc);
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (d >= g.rank) break;
    const double x = (double)xin[d];
    double s;
    if (in.range_mode == NUFFT_HIP_RANGE_STRICT) {
      *bad |= !(x > -kPiD && x < kPiD);   // IsWithinRange, nufft_plan.h:866-898 (strict inequalities)
      s = x + kPiD;
    } else if (in.range_mode == NUFFT_HIP_RANGE_EXTENDED) {
      *bad |= !(x > -3.0 * kPiD && x < 3.0 * kPiD);
      s = (x > kPiD) ? x - kPiD : ((x < -kPiD) ? x + 3.0 * kPiD : x + kPiD);
    } else {
      s = fmod(x + kPiD, 2.0 * kPiD);
      if (s < 0.0) s += 2.0 * kPiD;
    }
    const int nf = g.nf[d];
    double xs = s * ((double)nf * (1.0 / (2.0 * kPiD)));
    if (!(xs > -1.0e15 && xs < 1.0e15)) xs = 0.0;   // NaN / inf / absurd: keep memory safe
    const double i0f = ceil(xs - 0.5 * (double)g.w);
    double zz = 2.0 * (i0f - xs) + (double)(g.w - 1);
    zz = fmin(1.0, fmax(-1.0, zz));
    // periodic wrap of the stencil start: in-range points need at most one
    // add/subtract; the 64-bit modulo is kept for out-of-range garbage only
    int i0;
    if (i0f >= -(double)nf && i0f < 2.0 * (double)nf) {
      i0 = (int)i0f;
      if (i0 < 0) i0 += nf;
      else if (i0 >= nf) i0 -= nf;
    } else {
      long long m = (long long)i0f % nf;
      if (m < 0) m += nf;
      i0 = (int)m;
    }
    int t, l;
    if (g.tile_shift[d] >= 0) {           // power-of-two tile: shift / mask
      t = i0 >> g.tile_shift[d];
      l = i0 & (g.tile[d] - 1);
    } else {
      t = i0 / g.tile[d];
      l = i0 - t * g.tile[d];
    }
    tc[d] = t;
    loc |= (uint32_t)l << (10 * d);
    zz3[d] = (T)zz;
  }
  r->loc = loc;
  r->z0 = zz3[0];
  r->z1 = zz3[1];
  r->z2 = zz3[2];
  return tile_id(g, tc);
}

template <typename T>
__device__ __forceinline__ int fold_point(const Geom& g, const PointsIn& in, int64_t i, Rec<T>* r,
                                          bool* bad) {
  T x[3];
  load_coords<T, 0>(in, i, x);
  return fold_coords<T>(g, in, x, r, bad);
}

// Final form of a record: the point index joins it; 3-D float packs the Horner arguments.
template <typename T>
__device__ __forceinline__ Rec<T> pack_record(int rank, Rec<T> r, int32_t idx);
template <>
__device__ __forceinline__ Rec<float> pack_record<float>(int rank, Rec<float> r, int32_t idx) {
  if (rank < 3) {
    r.idx = idx;
  } else {
    // 3-D float: 28-bit fixed-point Horner arguments + 4-bit tile-local starts per
    // dimension + the point index, so the record stays ONE 16-byte store (a
    // separate 4-byte index array doubled the scattered write transactions)
    const uint32_t l0 = r.loc & 1023u, l1 = (r.loc >> 10) & 1023u, l2 = (r.loc >> 20) & 1023u;
    uint32_t w[3];
    const float z[3] = {r.z0, r.z1, r.z2};
    const uint32_t l[3] = {l0, l1, l2};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      float q = (z[d] + 1.0f) * 134217728.0f;   // 2^27
      q = fminf(fmaxf(q, 0.0f), 268435455.0f);
      w[d] = (uint32_t)__float2uint_rn(q);
      if (w[d] > 268435455u) w[d] = 268435455u;
      w[d] |= l[d] << 28;
    }
    r.loc = w[0];
    r.z0 = __uint_as_float(w[1]);
    r.z1 = __uint_as_float(w[2]);
    r.idx = idx;
  }
  return r;
}
template <>
__device__ __forceinline__ Rec<double> pack_record<double>(int rank, Rec<double> r, int32_t idx) {
  (void)rank;
  r.idx = idx;
  return r;
}
template <typename T>
__device__ __forceinline__ void store_record(const SortedOut<T>& out, int rank, int pos, Rec<T> r, int32_t idx) {
  out.rec[pos] = pack_record<T>(rank, r, idx);   // one 16-byte (float) / 32-byte (double) store
}

// --- path A (ntiles <= kMaxLdsTiles): counting sort with per-workgroup LDS
// histograms. No global atomics, deterministic tile order. Three passes over
// the raw points (hist, scatter) + two tiny scans.
constexpr int kSortBlock = 1024;
// Minimum waves per SIMD the count / scatter kernels are compiled for: 8 = two 1024-thread
// workgroups per CU (needs <= 64 VGPRs: the compiler spills 16-90 bytes per lane), 4 = one.
#ifndef NUFFT_SORT_MIN_WAVES
#define NUFFT_SORT_MIN_WAVES 8
#endif
// (the general fold -- fmod / division / 64-bit modulo paths -- and the double instantiations do not fit in 64 VGPRs:
// they are compiled for one workgroup per CU instead of spilling 24-92 bytes per lane; r04 verdict, weak #9)
template <typename T, bool QF> constexpr int sort_min_waves() { return (QF && sizeof(T) == 4) ? NUFFT_SORT_MIN_WAVES : 4; }
constexpr int kSortBatch = 4;    // load slots in flight per thread in the count / staged-scatter loops
// The plain scatter of the two-call form (one scattered 16-byte store per point, strengths not read) runs ahead of its
// stores with twice the loads in flight: r04 same-run A/B at config 2 (set_points + execute), slots 2 / 4 / 8: scatter
// 180 / 164 / 146-153 us (the count pass: 30 / 30 / 31 us, the 3-D count 272 / 259 / 309 us with spills: it keeps 4;
// the fused scatter of the one-call form: 166 / 167 us with 4 / 8).
constexpr int kScatterBatch = 8;

// Walks the points [lo, hi) of a workgroup, kSortBatch load slots per thread per pass, the
// loads of a pass issued back to back on clamped indices (with one load in flight per
// thread the kernels ran at the latency bound: 32 waves x 512 B per CU / ~2 us = 1.7 TB/s;
// a load under a divergent branch is waited for before the next one is issued).
// PAIR (float, [M, 2] interleaved points): a slot is ONE 16-byte load = two consecutive
// points (lo is even; an odd last point is handled by thread 0 after the loop), which
// halves the load instructions again. body(i, x[3], point_slot) is called for valid points
// only; pre(load_slot, i_clamped, valid) runs in the load phase for callers with a second
// array (load_slot = point_slot = -1 for the odd tail point).
// The points a sort workgroup handles: a contiguous range inside ONE point set (item).
struct BlockRange {
  int64_t lo, hi, base;   // global point indices; base = first point of the item
  int tile_off;           // item * ntiles_item
  __device__ __forceinline__ BlockRange(const PointsIn& in, int64_t per_block, const Geom& g) {
    const int item = (int)blockIdx.x / in.blocks_per_item;
    const int bi = (int)blockIdx.x - item * in.blocks_per_item;
    base = (int64_t)item * in.M_item;
    lo = base + (int64_t)bi * per_block;
    const int64_t end = base + in.M_item;
    if (lo > end) lo = end;
    hi = lo + per_block < end ? lo + per_block : end;
    tile_off = item * g.ntiles_item;
  }
};

template <typename T, int AOS, int BATCH = kSortBatch>
struct PointWalk {
  static constexpr bool PAIR = (AOS == 2 && sizeof(T) == 4);
  static constexpr int PP = PAIR ? 2 : 1;
  static constexpr int NS = PAIR ? BATCH / 2 : BATCH;   // load slots per pass: BATCH points per thread either way
  template <typename Pre, typename Body>
  static __device__ __forceinline__ void run(const PointsIn& in, int64_t lo, int64_t hi, Pre pre, Body body) {
    const int64_t hi_main = PAIR ? (hi & ~(int64_t)1) : hi;
    for (int64_t i0 = lo + PP * (int64_t)threadIdx.x; i0 < hi_main; i0 += (int64_t)PP * NS * kSortBlock) {
      T x[NS * PP][3];
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        const int64_t i = i0 + (int64_t)u * PP * kSortBlock;
        const int64_t ic = i < hi_main ? i : hi_main - PP;
        if constexpr (PAIR) {
          const float4 v = reinterpret_cast<const float4*>(in.pts[1])[ic >> 1];
          x[2 * u][0] = v.y; x[2 * u][1] = v.x; x[2 * u][2] = v.y;
          x[2 * u + 1][0] = v.w; x[2 * u + 1][1] = v.z; x[2 * u + 1][2] = v.w;
        } else {
          load_coords<T, AOS>(in, ic, x[u]);
        }
        pre(u, ic, i < hi_main);
      }
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        const int64_t i = i0 + (int64_t)u * PP * kSortBlock;
        if (i < hi_main) {
#pragma unroll
          for (int k = 0; k < PP; ++k) body(i + k, x[PP * u + k], PP * u + k);
        }
      }
    }
    if constexpr (PAIR) {
      if ((hi & 1) && threadIdx.x == 0) {   // odd tail point (last workgroup only)
        T x[3];
        load_coords<T, AOS>(in, hi - 1, x);
        pre(-1, hi - 1, true);   // slot -1 = the tail point
        body(hi - 1, x, -1);
      }
    }
  }
};

template <typename T, int AOS, bool QF = false>
__global__ __launch_bounds__(kSortBlock, (sort_min_waves<T, QF>())) void hist_lds_kernel(Geom g, PointsIn in, int64_t per_block,
                                                              int32_t* __restrict__ hist,
                                                              int32_t* __restrict__ bad_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int* h = reinterpret_cast<int*>(smem_raw);
  const int nt = g.ntiles;
  for (int t = threadIdx.x; t < nt; t += kSortBlock) h[t] = 0;
  __syncthreads();
  const BlockRange br(in, per_block, g);
  bool bad = false;
  PointWalk<T, AOS>::run(
      in, br.lo, br.hi, [](int, int64_t, bool) {},
      [&](int64_t, const T* x, int) {
        Rec<T> r;
        const int tile = fold_coords<T, QF>(g, in, x, &r, &bad) + br.tile_off;
        atomicAdd(&h[tile], 1);
      });
  if (bad && in.check_range) atomicAdd(bad_count, 1);
  __syncthreads();
  int32_t* out = hist + (int64_t)blockIdx.x * nt;
  for (int t = threadIdx.x; t < nt; t += kSortBlock) out[t] = h[t];
}

// hist[b][t] -> exclusive prefix over b (per tile), totals -> tile_count[t].
// A workgroup owns 16 tile columns (64-byte row segments; the table was just written and
// sits in L2) and splits the rows into 64 groups: thread (group, column) sums its rows,
// the 64 group sums of a column are scanned through LDS, then every thread rewrites its
// rows with the running prefix. 4x the workgroups and 1/4 of the serial rows per thread
// of the first version (64 columns x 16 bands): 16 -> ~8 us at config 2.
constexpr int kScanCols = 16, kScanGroups = 64, kScanRowsMax = 16;
__global__ __launch_bounds__(1024) void colscan_kernel(int nt, int nblk, int32_t* __restrict__ hist,
                                                       int32_t* __restrict__ tile_count) {
  __shared__ int part[kScanGroups][kScanCols + 1];
  const int tx = threadIdx.x & (kScanCols - 1), ty = threadIdx.x / kScanCols;
  const int t = blockIdx.x * kScanCols + tx;
  const int tc = t < nt ? t : nt - 1;
  const int rpg = (nblk + kScanGroups - 1) / kScanGroups;   // rows per group (<= kScanRowsMax)
  const int r0 = ty * rpg;
  int v[kScanRowsMax];
  int sum = 0;
#pragma unroll
  for (int k = 0; k < kScanRowsMax; ++k) {   // unconditional loads on clamped rows: all in flight together
    const int r = r0 + k;
    v[k] = hist[(int64_t)(r < nblk ? r : nblk - 1) * nt + tc];
    if (k >= rpg || r >= nblk) v[k] = 0;
    sum += v[k];
  }
  part[ty][tx] = sum;
  __syncthreads();
  int run = 0;
  for (int k = 0; k < ty; ++k) run += part[k][tx];
  if (t < nt) {
#pragma unroll
    for (int k = 0; k < kScanRowsMax; ++k) {
      const int r = r0 + k;
      if (k < rpg && r < nblk) hist[(int64_t)r * nt + t] = run;
      run += v[k];
    }
    if (ty == kScanGroups - 1) tile_count[t] = run;
  }
}

// FUSED (float, rank 2): the records carry the strengths (FusedRec) instead of the point
// index; in.strengths is read alongside the points, one 16-byte load per point pair.
__device__ __forceinline__ uint32_t fused_pack(uint32_t l, float z) {
  float q = (z + 1.0f) * kFusedScale;
  q = fminf(fmaxf(q, 0.0f), 134217727.0f);   // 2^27 - 1
  uint32_t w = (uint32_t)__float2uint_rn(q);
  if (w > 134217727u) w = 134217727u;
  return (l << 27) | w;
}
// signed conversion: the 24-bit rounding of the int -> float conversion then acts on |z| (as in a native float z)
__device__ __forceinline__ float fused_z(uint32_t p) { return (float)((int)(p & 0x7ffffffu) - (1 << 26)) * kFusedInv; }
__device__ __forceinline__ uint32_t fused_loc(uint32_t px, uint32_t py) { return (px >> 27) | ((py >> 27) << 10); }

__device__ __forceinline__ FusedRec fused_record(const Rec<float>& r, float2 cv) {
  FusedRec fr;
  fr.px = fused_pack(r.loc & 1023u, r.z0);
  fr.py = fused_pack((r.loc >> 10) & 1023u, r.z1);
  fr.re = cv.x;
  fr.im = cv.y;
  return fr;
}

template <typename T, int AOS, bool FUSED, bool QF = false>
__global__ __launch_bounds__(kSortBlock, (sort_min_waves<T, QF>())) void scatter_lds_kernel(Geom g, PointsIn in, int64_t per_block,
                                                                 const int32_t* __restrict__ hist,
                                                                 const int32_t* __restrict__ tile_start,
                                                                 SortedOut<T> out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int* cur = reinterpret_cast<int*>(smem_raw);
  const int nt = g.ntiles;
  const int32_t* hb = hist + (int64_t)blockIdx.x * nt;
  for (int t = threadIdx.x; t < nt; t += kSortBlock) cur[t] = tile_start[t] + hb[t];
  __syncthreads();
  const BlockRange br(in, per_block, g);
  const int64_t lo = br.lo, hi = br.hi;
  bool bad = false;
  // (float [M, 2] points with the short fold, unfused: 57 VGPRs, no scratch; the fused form already reads two arrays --
  // no gain measured, 20 bytes of scratch -- and the other forms would spill 50-200 bytes)
  using Walk = PointWalk<T, AOS, (sizeof(T) == 4 && AOS == 2 && QF && !FUSED ? kScatterBatch : kSortBatch)>;
  if constexpr (FUSED) {
    static_assert(sizeof(T) == 4, "fused records are float only");
    float2 cs[Walk::NS * Walk::PP];
    float2 ctail = make_float2(0.f, 0.f);
    Walk::run(
        in, lo, hi,
        [&](int u, int64_t ic, bool) {
          if constexpr (Walk::PAIR) {
            if (u >= 0) {
              const float4 v = reinterpret_cast<const float4*>(in.strengths)[ic >> 1];
              cs[2 * u] = make_float2(v.x, v.y);
              cs[2 * u + 1] = make_float2(v.z, v.w);
            } else {   // the odd tail point
              ctail = reinterpret_cast<const float2*>(in.strengths)[ic];
            }
          } else {
            cs[u] = reinterpret_cast<const float2*>(in.strengths)[ic];
          }
        },
        [&](int64_t, const T* x, int slot) {
          Rec<T> r;
          const int tile = fold_coords<T, QF>(g, in, x, &r, &bad) + br.tile_off;
          const float2 cv = slot >= 0 ? cs[slot] : ctail;
          const int pos = atomicAdd(&cur[tile], 1);
          reinterpret_cast<FusedRec*>(out.rec)[pos] = fused_record(r, cv);   // one 16-byte store
        });
  } else {
    Walk::run(
        in, lo, hi, [](int, int64_t, bool) {},
        [&](int64_t i, const T* x, int) {
          Rec<T> r;
          const int tile = fold_coords<T, QF>(g, in, x, &r, &bad) + br.tile_off;
          const int pos = atomicAdd(&cur[tile], 1);
          store_record<T>(out, g.rank, pos, r, (int32_t)(i - br.base));   // index inside the point set
        });
  }
}

// --- path A, staged scatter (at most kStagedMaxTiles tiles per point set: 2-D type-2 plans
// with 64 x 64 tiles, batches of 512^2-sized items, small grids). Scattered 16-byte stores are
// bound by write TRANSACTIONS, and records of one (workgroup, tile) run that arrive at
// different times are separate transactions (EXPERIMENTS.md section 5). Here a workgroup takes its
// points kStagedChunk at a time, orders the chunk by tile in LDS (counting sort: returning LDS
// atomics for the ranks, a 1024-entry scan) and writes every tile's records of the chunk with
// CONSECUTIVE LANES: ~8 records = 128 bytes per store group at 1024 tiles.
// Level-1 record: (6-bit super-tile-local start | 26-bit Horner argument) x 3 + the point index.
__device__ __forceinline__ uint32_t coarse_pack(uint32_t l, float z) {
  float q = (z + 1.0f) * 33554432.0f;   // 2^25
  q = fminf(fmaxf(q, 0.0f), 67108863.0f);
  uint32_t w = (uint32_t)__float2uint_rn(q);
  if (w > 67108863u) w = 67108863u;
  return (l << 26) | w;
}
constexpr int kStagedMaxTiles = 1024;
// (r03: 64 KB of staging, i.e. two workgroups per CU with 4-record runs, measured 100 -> 158 us at config 3)
constexpr int kStagedBytes = 128 * 1024;   // LDS for the staged records
// records staged per pass: NTMAX = 1024: 8192 float / 4096 double; NTMAX = 4096 (three 16 KB counter
// arrays instead of three 4 KB ones): 6144 / 3072
template <typename T, int NTMAX> constexpr int kStagedChunkOf =
    NTMAX == 1024 ? kStagedBytes / (int)sizeof(Rec<T>) : (sizeof(Rec<T>) == 16 ? 6144 : 3072);
template <typename T> constexpr int kStagedChunk = kStagedChunkOf<T, 1024>;
template <typename T, int NTMAX> constexpr size_t kStagedLds =
    (size_t)kStagedChunkOf<T, NTMAX> * (sizeof(Rec<T>) + 2) + 3 * NTMAX * 4 + 64;

// COARSE (3-D float, level 1 of the two-level sort below): g is the coarse geometry (tiles = super-tiles) and the
// records take the level-1 form (coarse_pack).
template <typename T, int AOS, bool FUSED, int NTMAX = 1024, bool COARSE = false, bool QF = false>
__global__ __launch_bounds__(kSortBlock, 4) void scatter_staged_kernel(Geom g, PointsIn in, int64_t per_block,
                                                                      const int32_t* __restrict__ hist,
                                                                      const int32_t* __restrict__ tile_start,
                                                                      SortedOut<T> out) {
  using RecT = std::conditional_t<FUSED, FusedRec, Rec<T>>;
  static_assert(sizeof(RecT) == sizeof(Rec<T>), "record sizes");
  constexpr int CHUNK = kStagedChunkOf<T, NTMAX>;
  constexpr int PER = CHUNK / kSortBlock;            // points per thread and chunk
  constexpr int K = NTMAX / kSortBlock;              // tile counters per thread
  constexpr int TB = NTMAX == 1024 ? 10 : 12;        // bits of the tile in the (tile, rank) word
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  RecT* stage = reinterpret_cast<RecT*>(smem_raw);                       // [CHUNK]
  uint16_t* tl = reinterpret_cast<uint16_t*>(stage + CHUNK);             // [CHUNK] tile of the staged record
  int* cur = reinterpret_cast<int*>(tl + CHUNK);                         // [NTMAX] global cursor
  int* cnt = cur + NTMAX;                                                // records of the chunk per tile
  int* off = cnt + NTMAX;                                                // their exclusive scan
  int* wsum = off + NTMAX;                                               // [16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ntl = g.ntiles_item;
  const BlockRange br(in, per_block, g);
  const int32_t* hb = hist + (int64_t)blockIdx.x * g.ntiles + br.tile_off;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int t = tid * K + k;
    cur[t] = t < ntl ? tile_start[br.tile_off + t] + hb[t] : 0;
    cnt[t] = 0;
  }
  __syncthreads();
  bool bad = false;
  for (int64_t cb = br.lo; cb < br.hi; cb += CHUNK) {
    const int64_t ce = cb + CHUNK < br.hi ? cb + CHUNK : br.hi;
    // ---- phase 1: load, fold, count (loads of a thread issued back to back on clamped indices;
    // fetching the next chunk during phases 2-5 was tried: it spills 90-160 bytes per lane and
    // gains nothing, 101 -> 98 us at config 3)
    T x[PER][3];
    float2 cs[FUSED ? PER : 1];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int64_t i = cb + (int64_t)u * kSortBlock + tid;
      const int64_t ic = i < ce ? i : ce - 1;
      load_coords<T, AOS>(in, ic, x[u]);
      if constexpr (FUSED) cs[u] = reinterpret_cast<const float2*>(in.strengths)[ic];
    }
    RecT rec[PER];
    int tr[PER];     // tile | rank << TB, -1 past the end
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int64_t i = cb + (int64_t)u * kSortBlock + tid;
      Rec<T> r;
      const int tile = fold_coords<T, QF>(g, in, x[u], &r, &bad);
      tr[u] = -1;
      if (i < ce) tr[u] = tile | (atomicAdd(&cnt[tile], 1) << TB);
      if constexpr (FUSED) {
        rec[u] = fused_record(r, cs[u]);
      } else if constexpr (COARSE) {
        rec[u].loc = coarse_pack(r.loc & 1023u, r.z0);
        rec[u].z0 = __uint_as_float(coarse_pack((r.loc >> 10) & 1023u, r.z1));
        rec[u].z1 = __uint_as_float(coarse_pack((r.loc >> 20) & 1023u, r.z2));
        rec[u].idx = (int32_t)(i - br.base);
      } else {
        rec[u] = pack_record<T>(g.rank, r, (int32_t)(i - br.base));
      }
    }
    __syncthreads();
    // ---- phase 2: exclusive scan of the chunk's tile counts (K consecutive entries per thread)
    {
      int v[K], tot = 0;
#pragma unroll
      for (int k = 0; k < K; ++k) { v[k] = cnt[tid * K + k]; tot += v[k]; }
      int incl = tot;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
      }
      if (lane == 63) wsum[wave] = incl;
      __syncthreads();
      int run = incl - tot;
      for (int k = 0; k < wave; ++k) run += wsum[k];
#pragma unroll
      for (int k = 0; k < K; ++k) { off[tid * K + k] = run; run += v[k]; }
    }
    __syncthreads();
    // ---- phase 3: records to their place in the chunk's tile order
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      if (tr[u] >= 0) {
        const int tile = tr[u] & ((1 << TB) - 1);
        const int sidx = off[tile] + (tr[u] >> TB);
        stage[sidx] = rec[u];
        tl[sidx] = (uint16_t)tile;
      }
    }
    __syncthreads();
    // ---- phase 4: write out, consecutive lanes = consecutive records of a tile
    const int nchunk = (int)(ce - cb);
    RecT* dst = reinterpret_cast<RecT*>(out.rec);
    for (int sidx = tid; sidx < nchunk; sidx += kSortBlock) {
      const int tile = tl[sidx];
      dst[cur[tile] + sidx - off[tile]] = stage[sidx];
    }
    __syncthreads();
    // ---- phase 5: advance the cursors
#pragma unroll
    for (int k = 0; k < K; ++k) {
      cur[tid * K + k] += cnt[tid * K + k];
      cnt[tid * K + k] = 0;
    }
    __syncthreads();
  }
}

// --- path A16 (16384 < ntiles <= kMaxRanges16 * kMaxLds16Tiles): same scheme with
// 16-bit counters packed two per LDS word, so that 65536 tiles (3-D 512^3 at 16x16x8)
// fit in 128 KiB of LDS. A workgroup never takes more than 65535 points, so a
// packed counter cannot carry into its neighbour. The per-(workgroup, tile)
// prefix is 32-bit and lives in HBM; the scatter keeps only RELATIVE 16-bit
// cursors in LDS and adds tile_start + prefix read from L2. More tiles than fit LDS
// (3-D at w = 7, 8: 16x16x4 tiles, 131072 of them on 512^3) are covered by up to
// kMaxRanges16 tile RANGES: blockIdx.y picks the range a workgroup counts, every range
// re-reads the block's points (count 1.33 -> 0.35 ms at M = 3e7 against the global-counter
// path that served these geometries before).
template <typename T>
__global__ __launch_bounds__(kSortBlock) void hist16_lds_kernel(Geom g, PointsIn in, int64_t per_block,
                                                                int span, uint32_t* __restrict__ hist16,
                                                                uint16_t* __restrict__ rank16,
                                                                int32_t* __restrict__ bad_count) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned* h2 = reinterpret_cast<unsigned*>(smem_raw);
  const int nw = (g.ntiles + 1) >> 1;
  const int t_lo = (int)blockIdx.y * span;                                   // span is even
  const int t_hi = (t_lo + span < g.ntiles) ? t_lo + span : g.ntiles;
  const int nwr = (t_hi - t_lo + 1) >> 1;                                    // words of this range
  for (int t = threadIdx.x; t < nwr; t += kSortBlock) h2[t] = 0u;
  __syncthreads();
  const BlockRange br(in, per_block, g);
  const int64_t lo = br.lo, hi = br.hi;
  bool bad = false;
  for (int64_t i0 = lo + threadIdx.x; i0 < hi; i0 += kSortBatch * kSortBlock) {
    T x[kSortBatch][3];
#pragma unroll
    for (int u = 0; u < kSortBatch; ++u) {
      const int64_t i = i0 + u * kSortBlock;
      load_coords<T, 0>(in, i < hi ? i : hi - 1, x[u]);
    }
#pragma unroll
    for (int u = 0; u < kSortBatch; ++u) {
      const int64_t i = i0 + u * kSortBlock;
      Rec<T> r;
      const int tile = fold_coords<T>(g, in, x[u], &r, &bad) + br.tile_off;
      if (i < hi && tile >= t_lo && tile < t_hi) {
        const int tl = tile - t_lo;
        const int sh = 16 * (tl & 1);
        const unsigned old = atomicAdd(&h2[tl >> 1], 1u << sh);
        // (the scatter pass folds the point again and gets the same tile: no per-point tile array on this path --
        // 0.4 GB less written here and read there at M = 1e8)
        rank16[i] = (uint16_t)((old >> sh) & 0xffffu);   // rank inside (workgroup, tile)
      }
    }
  }
  if (bad && in.check_range && blockIdx.y == 0) atomicAdd(bad_count, 1);
  __syncthreads();
  uint32_t* out = hist16 + (int64_t)blockIdx.x * nw + (t_lo >> 1);
  for (int t = threadIdx.x; t < nwr; t += kSortBlock) out[t] = h2[t];
}

// hist16[b][t] (uint16) -> pref[b][t] (int32 exclusive prefix over b), totals -> tile_count
__global__ __launch_bounds__(1024) void colscan16_kernel(int nt, int nblk, const uint16_t* __restrict__ hist16,
                                                         int32_t* __restrict__ pref,
                                                         int32_t* __restrict__ tile_count) {
  __shared__ int part[16][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int t = blockIdx.x * 64 + tx;
  const int nt2 = ((nt + 1) >> 1) << 1;   // row pitch of hist16 in uint16
  const int rpg = (nblk + 15) / 16;
  const int r0 = ty * rpg;
  const int r1 = (r0 + rpg < nblk) ? r0 + rpg : nblk;
  int sum = 0;
  if (t < nt)
    for (int r = r0; r < r1; ++r) sum += hist16[(int64_t)r * nt2 + t];
  part[ty][tx] = sum;
  __syncthreads();
  int run = 0;
  for (int k = 0; k < ty; ++k) run += part[k][tx];
  if (t < nt) {
    for (int r = r0; r < r1; ++r) {
      const int v = hist16[(int64_t)r * nt2 + t];
      pref[(int64_t)r * nt + t] = run;
      run += v;
    }
    if (ty == 15) tile_count[t] = run;
  }
}

// Streaming scatter for path A16: position = tile_start + prefix of the point's
// workgroup + its 16-bit rank; no LDS, full occupancy.
template <typename T, bool FUSED3 = false>
__global__ __launch_bounds__(256) void scatter_ranked_kernel(Geom g, PointsIn in, int64_t per_block,
                                                             const uint16_t* __restrict__ rank16,
                                                             const int32_t* __restrict__ pref,
                                                             const int32_t* __restrict__ tile_start,
                                                             SortedOut<T> out) {
  const int64_t il = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // index inside the point set blockIdx.y
  if constexpr (FUSED3) {
    // record + strength = 32 bytes per point. A lane stores at most 16 bytes per instruction, and two 16-byte
    // stores per lane are two write transactions per point (measured r03: scatter 3.1 -> 4.2 ms at M = 1e8).
    // So lane PAIRS store: instruction 1 writes the even lane's record (even lane the first half, odd lane the
    // second), instruction 2 the odd lane's -- every instruction then touches 32 whole 32-byte blocks.
    const bool live = il < in.M_item;
    const int64_t ic = live ? il : in.M_item - 1;                        // (clamped: every lane takes part in the exchange)
    const int64_t i = (int64_t)blockIdx.y * in.M_item + ic;
    Rec<T> r;
    bool bad = false;
    const int tile = fold_point<T>(g, in, i, &r, &bad) + (int)blockIdx.y * g.ntiles_item;
    const int64_t blk = (int64_t)blockIdx.y * in.blocks_per_item + ic / per_block;
    const int pos = live ? tile_start[tile] + pref[blk * g.ntiles + tile] + (int)rank16[i] : -1;
    const float2 cv = reinterpret_cast<const float2*>(in.strengths)[i];
    const Rec<float> pr = pack_record<float>(3, r, (int32_t)ic);
    const uint4 lo = {pr.loc, __float_as_uint(pr.z0), __float_as_uint(pr.z1), (uint32_t)pr.idx};
    const uint4 hi = {__float_as_uint(cv.x), __float_as_uint(cv.y), 0u, 0u};
    const int lane = threadIdx.x & 63, odd = lane & 1;
    uint4* dst = reinterpret_cast<uint4*>(out.rec);
    // what the partner lane (lane ^ 1) holds
    const int ppos = __shfl_xor(pos, 1);
    uint4 plo, phi;
    plo.x = __shfl_xor(lo.x, 1); plo.y = __shfl_xor(lo.y, 1); plo.z = __shfl_xor(lo.z, 1); plo.w = __shfl_xor(lo.w, 1);
    phi.x = __shfl_xor(hi.x, 1); phi.y = __shfl_xor(hi.y, 1); phi.z = __shfl_xor(hi.z, 1); phi.w = __shfl_xor(hi.w, 1);
    // instruction 1: the even lane's record; instruction 2: the odd lane's
    const int pa = odd ? ppos : pos;
    const uint4 va = odd ? phi : lo;
    if (pa >= 0) dst[2 * (int64_t)pa + odd] = va;
    const int pb = odd ? pos : ppos;
    const uint4 vb = odd ? hi : plo;
    if (pb >= 0) dst[2 * (int64_t)pb + odd] = vb;
    return;
  }
  if (il >= in.M_item) return;
  const int64_t i = (int64_t)blockIdx.y * in.M_item + il;
  Rec<T> r;
  bool bad = false;
  const int tile = fold_point<T>(g, in, i, &r, &bad) + (int)blockIdx.y * g.ntiles_item;
  const int64_t blk = (int64_t)blockIdx.y * in.blocks_per_item + il / per_block;
  const int pos = tile_start[tile] + pref[blk * g.ntiles + tile] + (int)rank16[i];
  store_record<T>(out, g.rank, pos, r, (int32_t)il);
}

// --- path S (two levels; 3-D float plans whose tiles are numbered by super-tiles, Geom::sup_shift).
// The one-pass scatters above write every record to a random place: one 32-byte write transaction per
// record, ~3e10 per second whatever the record size, and the 16-bit path adds a random 4-byte read of its
// [workgroup][tile] prefix table per point (a 64-byte sector each; EXPERIMENTS.md section 5). Here the points go
// to their super-tile first (64^3 fine cells; <= 1024 destinations, so a workgroup's 8192-point pass writes
// ~16 records = 256 bytes per destination with consecutive lanes), and then every <= 4096-record piece of a
// super-tile is ordered by tile inside it (<= 256 keys: segments of ~32-64 records). Neither level stages
// records in LDS: a workgroup computes the permutation of its piece (LDS counters, 16-bit indices) and then
// walks the OUTPUT positions, re-reading its input through the permutation (it was read a moment ago: L2).
constexpr int kSort2Sub = 4096;         // records per level-2 workgroup (a piece of a super-tile): 64 KB staged
constexpr size_t kSort2Lds = (size_t)kSort2Sub * 17 + 2 * 256 * 4 + 16;
constexpr int kSort2Threads = 512;      // threads of a level-2 count workgroup
constexpr int kSort2MaxKeys = 256;      // tiles per super-tile

// tile of a level-1 record inside its super-tile (fine geometry g), and the record's final form
__device__ __forceinline__ int coarse_key(const Geom& g, const uint4& r) {
  const int k0 = (int)(r.x >> 26) >> g.tile_shift[0], k1 = (int)(r.y >> 26) >> g.tile_shift[1],
            k2 = (int)(r.z >> 26) >> g.tile_shift[2];
  return k0 | (k1 << g.sup_shift[0]) | (k2 << (g.sup_shift[0] + g.sup_shift[1]));
}
__device__ __forceinline__ uint4 coarse_to_final(const Geom& g, const uint4& r) {
  uint4 o;
  o.x = (((r.x >> 26) & (uint32_t)(g.tile[0] - 1)) << 28) | ((r.x & 0x3ffffffu) << 2);
  o.y = (((r.y >> 26) & (uint32_t)(g.tile[1] - 1)) << 28) | ((r.y & 0x3ffffffu) << 2);
  o.z = (((r.z >> 26) & (uint32_t)(g.tile[2] - 1)) << 28) | ((r.z & 0x3ffffffu) << 2);
  o.w = r.w;
  return o;
}

// Level 2, count: workgroup = one piece (<= kSort2Sub records) of a super-tile; hist2[piece][key].
__global__ __launch_bounds__(kSort2Threads) void count2_kernel(Geom g, Geom g1, const int32_t* __restrict__ c_start,
                                                               const int32_t* __restrict__ c_sub,
                                                               const uint4* __restrict__ tmp_rec,
                                                               int32_t* __restrict__ hist2, int nkeys) {
  __shared__ int cnt[kSort2MaxKeys];
  int st, p0, p1, slot;
  if (!locate_subproblem(g1, c_start, c_sub, blockIdx.x, &st, &p0, &p1, &slot)) return;
  const int tid = threadIdx.x;
  if (tid < kSort2MaxKeys) cnt[tid] = 0;
  __syncthreads();
  constexpr int PER = kSort2Sub / kSort2Threads;
  const int n = p1 - p0;
  uint4 r[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = tid + u * kSort2Threads;
    if (u * kSort2Threads < n) r[u] = tmp_rec[p0 + (i < n ? i : n - 1)];
  }
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = tid + u * kSort2Threads;
    if (i < n) atomicAdd(&cnt[coarse_key(g, r[u])], 1);
  }
  __syncthreads();
  if (tid < nkeys) hist2[(int64_t)blockIdx.x * nkeys + tid] = cnt[tid];
}

// Level 2, scan: workgroup = super-tile. hist2[piece][key] -> exclusive prefix over the super-tile's pieces,
// totals -> tile_count[super-tile << bits | key].
__global__ __launch_bounds__(1024) void scan2_kernel(const int32_t* __restrict__ c_sub, int32_t* __restrict__ hist2,
                                                     int nkeys, int key_bits, int32_t* __restrict__ tile_count) {
  __shared__ int part[1024];
  const int st = blockIdx.x;
  const int r_lo = c_sub[st], r_hi = c_sub[st + 1];
  const int ngrp = 1024 / nkeys;
  const int key = threadIdx.x & (nkeys - 1), grp = threadIdx.x / nkeys;
  const int rpg = (r_hi - r_lo + ngrp - 1) / ngrp;
  const int r0 = r_lo + grp * rpg < r_hi ? r_lo + grp * rpg : r_hi;
  const int r1 = r0 + rpg < r_hi ? r0 + rpg : r_hi;
  int sum = 0;
  for (int r = r0; r < r1; ++r) sum += hist2[(int64_t)r * nkeys + key];
  part[threadIdx.x] = sum;
  __syncthreads();
  int run = 0, total = 0;
  for (int k = 0; k < ngrp; ++k) {
    const int v = part[k * nkeys + key];
    if (k < grp) run += v;
    total += v;
  }
  for (int r = r0; r < r1; ++r) {
    const int64_t at = (int64_t)r * nkeys + key;
    const int v = hist2[at];
    hist2[at] = run;
    run += v;
  }
  if (grp == 0) tile_count[(st << key_bits) | key] = total;
}

// Level 2, scatter: the piece's records to their tiles, in the final record form. The piece is ordered in LDS (a
// first version re-read its records from memory through a 16-bit permutation: the pieces in flight are twice
// the L2, every 16-byte gather then pulled a whole line back in -- 1.17 ms at M = 1e8 against 0.9 for this form).
__global__ __launch_bounds__(kSortBlock) void scatter2_kernel(Geom g, Geom g1, const int32_t* __restrict__ c_start,
                                                              const int32_t* __restrict__ c_sub,
                                                              const uint4* __restrict__ tmp_rec,
                                                              const int32_t* __restrict__ hist2, int nkeys, int key_bits,
                                                              const int32_t* __restrict__ tile_start,
                                                              uint4* __restrict__ out) {
  constexpr int NT = kSortBlock;
  constexpr int SUB = kSort2Sub;
  constexpr int PER = SUB / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint4* stage = reinterpret_cast<uint4*>(smem_raw);                 // [SUB]
  int* cnt = reinterpret_cast<int*>(stage + SUB);                   // [256]
  int* base = cnt + kSort2MaxKeys;                                  // [256]
  int* wsum = base + kSort2MaxKeys;                                 // [4]
  uint8_t* keyof = reinterpret_cast<uint8_t*>(wsum + 4);            // [SUB] key of the staged record
  int st, p0, p1, slot;
  if (!locate_subproblem(g1, c_start, c_sub, blockIdx.x, &st, &p0, &p1, &slot)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = p1 - p0;
  if (tid < kSort2MaxKeys) cnt[tid] = 0;
  __syncthreads();
  uint4 r[PER];
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = tid + u * NT;
    if (u * NT < n) r[u] = tmp_rec[p0 + (i < n ? i : n - 1)];
  }
  int kr[PER];   // key | rank << 8
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    const int i = tid + u * NT;
    kr[u] = -1;
    if (i < n) {
      const int key = coarse_key(g, r[u]);
      kr[u] = key | (atomicAdd(&cnt[key], 1) << 8);
    }
  }
  __syncthreads();
  if (tid < kSort2MaxKeys) {   // exclusive scan of the 256 counters by the first four waves
    const int v = cnt[tid];
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d);
      if (lane >= d) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    cnt[tid] = incl - v;   // (completed below)
  }
  __syncthreads();
  if (tid < kSort2MaxKeys) {
    int run = cnt[tid];
    for (int k = 0; k < wave; ++k) run += wsum[k];
    cnt[tid] = run;
    // output slot of the key's first record of this piece, minus its slot in the piece's key order
    base[tid] = tid < nkeys ? tile_start[(st << key_bits) | tid] + hist2[(int64_t)blockIdx.x * nkeys + tid] - run : 0;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < PER; ++u) {
    if (kr[u] >= 0) {
      const int at = cnt[kr[u] & 255] + (kr[u] >> 8);
      stage[at] = coarse_to_final(g, r[u]);
      keyof[at] = (uint8_t)(kr[u] & 255);
    }
  }
  __syncthreads();
  for (int sidx = tid; sidx < n; sidx += NT) {
    const int dst = base[keyof[sidx]] + sidx;
    out[dst] = stage[sidx];
  }
}

// --- path B (many tiles): per-point rank from a global counter (the
// reference's scheme, nufft_plan.cu.cc:160-296), then scatter.
template <typename T>
__global__ __launch_bounds__(256) void count_global_kernel(Geom g, PointsIn in,
                                                           int32_t* __restrict__ tile_of,
                                                           int32_t* __restrict__ rank_of,
                                                           int32_t* __restrict__ tile_count,
                                                           int32_t* __restrict__ bad_count) {
  const int64_t il = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (il >= in.M_item) return;
  const int64_t i = (int64_t)blockIdx.y * in.M_item + il;
  Rec<T> r;
  bool bad = false;
  const int tile = fold_point<T>(g, in, i, &r, &bad) + (int)blockIdx.y * g.ntiles_item;
  tile_of[i] = tile;
  rank_of[i] = atomicAdd(&tile_count[tile], 1);
  if (bad && in.check_range) atomicAdd(bad_count, 1);
}

template <typename T>
__global__ __launch_bounds__(256) void scatter_global_kernel(Geom g, PointsIn in,
                                                             const int32_t* __restrict__ tile_of,
                                                             const int32_t* __restrict__ rank_of,
                                                             const int32_t* __restrict__ tile_start,
                                                             SortedOut<T> out) {
  const int64_t il = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (il >= in.M_item) return;
  const int64_t i = (int64_t)blockIdx.y * in.M_item + il;
  Rec<T> r;
  bool bad = false;
  fold_point<T>(g, in, i, &r, &bad);
  const int pos = tile_start[tile_of[i]] + rank_of[i];
  store_record<T>(out, g.rank, pos, r, (int32_t)il);
}

// Single-workgroup exclusive scans over the tiles: point offsets and
// subproblem offsets (ceil(count / max_sub) per tile).
// sub_small > 0 (2-D type-2 plans on 64 x 64 tiles): the cap on the points of a subproblem is chosen
// HERE, from the tile counts. The interp kernel's workgroups are one subproblem long and the chip
// holds ~1024 of them at a time: a uniform point set is served best by ONE subproblem per tile
// (config 3: 1024 tiles of 9766 points, 262 us), but with a clustered one (radial trajectories:
// 218000 points in the densest tile) the full-cap workgroups of the dense tiles decide when the
// kernel ends (measured 430 us; a scheduling simulation of the measured tile counts reproduces
// it), while caps of 2048-4096 pack well (296-333 us) and cost the uniform case 20 %.
__global__ __launch_bounds__(1024) void scan_tiles_kernel(const int32_t* __restrict__ count, int n,
                                                          int max_sub, int sub_small, int avg,
                                                          int32_t* __restrict__ tile_start,
                                                          int32_t* __restrict__ sub_start) {
  __shared__ int ws_m[16];
  if (sub_small > 0) {
    int m = 0;
    for (int i = threadIdx.x; i < n; i += 1024) m = max(m, count[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_down(m, o));
    if ((threadIdx.x & 63) == 0) ws_m[threadIdx.x >> 6] = m;
    __syncthreads();
    m = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) m = max(m, ws_m[k]);
    if (2 * (long long)m > 3 * (long long)avg && sub_small < max_sub) max_sub = sub_small;
  }
  // One workgroup walks the tiles in blocks of 4096 (4 consecutive tiles per thread, so the
  // loads and stores of a wave are contiguous), scanning each block with wave shuffles and
  // carrying the running totals. (A fixed contiguous range per thread made every access of
  // a wave strided: 135 us for the 65536 tiles of config 4, 25 us this way.)
  __shared__ int ws_a[16], ws_b[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int run_a = 0, run_b = 0;
  int most = 0;   // most subproblems of any tile -> tile_start[n + 1] (fixed-point plans: is any tile crowded?)
  for (int base = 0; base < n; base += 4096) {
    const int i0 = base + 4 * tid;
    int c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = (i0 + k < n) ? count[i0 + k] : 0;
    int sa = 0, sb = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int ns = (c[k] + max_sub - 1) / max_sub;
      sa += c[k];
      sb += ns;
      most = max(most, ns);
    }
    int ia = sa, ib = sb;   // inclusive scans inside the wave, then across the 16 waves
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int ta = __shfl_up(ia, d), tb = __shfl_up(ib, d);
      if (lane >= d) { ia += ta; ib += tb; }
    }
    __syncthreads();   // previous block's readers of ws_* are done
    if (lane == 63) { ws_a[wave] = ia; ws_b[wave] = ib; }
    __syncthreads();
    int ea = run_a + ia - sa, eb = run_b + ib - sb, tot_a = 0, tot_b = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (k < wave) { ea += ws_a[k]; eb += ws_b[k]; }
      tot_a += ws_a[k];
      tot_b += ws_b[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (i0 + k < n) {
        tile_start[i0 + k] = ea;
        sub_start[i0 + k] = eb;
      }
      ea += c[k];
      eb += (c[k] + max_sub - 1) / max_sub;
    }
    run_a += tot_a;
    run_b += tot_b;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) most = max(most, __shfl_down(most, o));
  __syncthreads();
  if (lane == 0) ws_a[wave] = most;
  __syncthreads();
  if (tid == 0) {
    tile_start[n] = run_a;
    sub_start[n] = run_b;
    for (int k = 0; k < 16; ++k) most = max(most, ws_a[k]);
    tile_start[n + 1] = most;
  }
}

// ------------------------------------------------- spread: generic tile path

// One workgroup per subproblem (<= max_sub points of one tile) and transform.
// LDS tile of (tile + w - 1)^rank interleaved complex cells; one thread per
// point, w^rank LDS atomic adds per component; then the tile is added to the
// periodic fine grid with global float atomics (executed at the memory side
// on gfx950). Works for any w <= 16, rank, precision.
//
// The LDS tile is DOUBLE for both precisions: measured on MI355X
// (tools/ubench/lds_atomic_bench.hip, profiles/r01_lds_atomic_ubench.txt)
// ds_add_f32 is serialised at ~193 cycles per wave-instruction per CU
// whatever the occupancy, while ds_add_f64 takes ~8.6 -- 22x faster -- so
// fp32 strengths are accumulated in fp64 and rounded once on the way out.
template <typename T, int RANK>
__global__ __launch_bounds__(kBlock) void spread_tile_generic_kernel(
    Geom g, SortedPoints<T> sp, const T* __restrict__ horner, const T* __restrict__ c,
    T* __restrict__ fw, int64_t c_stride, int64_t fw_stride, T scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* tile = reinterpret_cast<double*>(smem_raw);
  int tb, p0, p1, slot;
  if (!locate_subproblem(g, sp.tile_start, sp.sub_start, blockIdx.x, &tb, &p0, &p1, &slot)) return;
  const int tid = threadIdx.x;
  const int w = g.w, nc = g.ncoef;
  const int L0 = g.ldim[0], LS = g.lstride;
  const int L1 = RANK > 1 ? g.ldim[1] : 1;
  const int L2 = RANK > 2 ? g.ldim[2] : 1;
  const int ncell_padded = LS * L1 * L2;
  for (int i = tid; i < 2 * ncell_padded; i += kBlock) tile[i] = 0.0;
  __syncthreads();

  const T* cc = c + 2 * (int64_t)slot * c_stride;
  for (int j = p0 + tid; j < p1; j += kBlock) {
    const PointView<T> rec = unpack_rec<T, RANK>(sp.rec[j]);
    const uint32_t loc = rec.loc;
    const int idx = rec.idx;
    const T re = cc[2 * (int64_t)idx] * scale;
    const T im = cc[2 * (int64_t)idx + 1] * scale;
    T kx[kMaxW];
    const T z0 = rec.z0;
#pragma unroll
    for (int q = 0; q < kMaxW; ++q) kx[q] = (q < w) ? horner_cell(horner, nc, q, z0) : (T)0;
    const int l0 = loc & 1023;
    if (RANK == 1) {
#pragma unroll
      for (int q = 0; q < kMaxW; ++q)
        if (q < w) {
          lds_add(&tile[2 * (l0 + q)], (double)(re * kx[q]));
          lds_add(&tile[2 * (l0 + q) + 1], (double)(im * kx[q]));
        }
    } else if (RANK == 2) {
      const int l1 = (loc >> 10) & 1023;
      const T z1 = rec.z1;
      for (int dy = 0; dy < w; ++dy) {
        const T ky = horner_cell(horner, nc, dy, z1);
        const T vre = re * ky, vim = im * ky;
        double* row = tile + 2 * ((l1 + dy) * LS + l0);
#pragma unroll
        for (int q = 0; q < kMaxW; ++q)
          if (q < w) {
            lds_add(&row[2 * q], (double)(vre * kx[q]));
            lds_add(&row[2 * q + 1], (double)(vim * kx[q]));
          }
      }
    } else {
      const int l1 = (loc >> 10) & 1023;
      const int l2 = (loc >> 20) & 1023;
      const T z1 = rec.z1;
      const T z2 = rec.z2;
      for (int dz = 0; dz < w; ++dz) {
        const T kz = horner_cell(horner, nc, dz, z2);
        for (int dy = 0; dy < w; ++dy) {
          const T kyz = kz * horner_cell(horner, nc, dy, z1);
          const T vre = re * kyz, vim = im * kyz;
          double* row = tile + 2 * (((l2 + dz) * L1 + (l1 + dy)) * LS + l0);
#pragma unroll
          for (int q = 0; q < kMaxW; ++q)
            if (q < w) {
              lds_add(&row[2 * q], (double)(vre * kx[q]));
              lds_add(&row[2 * q + 1], (double)(vim * kx[q]));
            }
        }
      }
    }
  }
  __syncthreads();

  // tile -> periodic fine grid
  int t0, t1, t2;
  tile_coords(g, tb, &t0, &t1, &t2);
  const int o0 = t0 * g.tile[0], o1 = t1 * g.tile[1], o2 = t2 * g.tile[2];
  T* out = fw + 2 * (int64_t)slot * fw_stride;
  const int ncell = L0 * L1 * L2;
  for (int i = tid; i < ncell; i += kBlock) {
    const int a0 = i % L0;
    const int a1 = (i / L0) % L1;
    const int a2 = i / (L0 * L1);
    const int li = (a2 * L1 + a1) * LS + a0;
    const T vre = (T)tile[2 * li], vim = (T)tile[2 * li + 1];
    if (vre != (T)0 || vim != (T)0) {
      const int64_t g0 = (o0 + a0) % g.nf[0];
      const int64_t g1 = RANK > 1 ? (o1 + a1) % g.nf[1] : 0;
      const int64_t g2 = RANK > 2 ? (o2 + a2) % g.nf[2] : 0;
      const int64_t gi = g0 + (int64_t)g.nf[0] * (g1 + (int64_t)g.nf[1] * g2);
      glb_add(&out[2 * gi], vre);
      glb_add(&out[2 * gi + 1], vim);
    }
  }
}

// ------------------------------------------- spread: sparse point sets, no LDS tile

// The LDS-tile kernels zero-fill and write out a whole tile (25-52 KB of LDS, tile + halo
// cells of global atomics) for every non-empty tile; below a few points per tile that
// overhead is all there is. Here every point adds its stencil straight to the fine grid
// (the reference's nupts-driven method, SpreadNuptsDriven*, nufft_plan.cu.cc:474-527,
// 707-787, 1190-1292, launched by spread_batch_nupts_driven :2325-2436), but shaped for
// the memory-side atomic units of gfx950: a stencil ROW is handled by 2 w consecutive lanes
// carrying (re, im) of consecutive cells, i.e. one contiguous 8 w-byte segment per row (the
// reference gives a thread a whole point and walks its w^d cells one 4-byte atomic at a
// time). One workgroup per subproblem of the tile-sorted records, as everywhere else, so
// empty tiles cost one early exit; the four waves take the subproblem's points in turn.
template <typename T, int RANK>
__global__ __launch_bounds__(kBlock) void spread_sparse_kernel(
    Geom g, SortedPoints<T> sp, const T* __restrict__ horner, const T* __restrict__ c,
    T* __restrict__ fw, int64_t c_stride, int64_t fw_stride, T scale) {
  int tb, p0, p1, slot;
  if (!locate_subproblem(g, sp.tile_start, sp.sub_start, blockIdx.x, &tb, &p0, &p1, &slot)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int w = g.w, nc = g.ncoef;
  // lanes per stencil row: 2 w rounded up to a power of two (4 .. 32)
  const int seg = w <= 2 ? 4 : w <= 4 ? 8 : w <= 8 ? 16 : 32;
  const int rsub = lane / seg, e = lane - rsub * seg;
  const int rows_per_pass = 64 / seg;
  const int dx = e >> 1, comp = e & 1;
  const bool lane_on = e < 2 * w;
  const int nrows = RANK == 1 ? 1 : RANK == 2 ? w : w * w;
  int t0, t1, t2;
  tile_coords(g, tb, &t0, &t1, &t2);
  const int o0 = t0 * g.tile[0], o1 = t1 * g.tile[1], o2 = t2 * g.tile[2];
  const T* cc = c + 2 * (int64_t)slot * c_stride;
  T* out = fw + 2 * (int64_t)slot * fw_stride;
  for (int j = p0 + wave; j < p1; j += kBlock / 64) {
    const PointView<T> rec = unpack_rec<T, RANK>(sp.rec[j]);   // wave-uniform address: one broadcast load
    const T cv = cc[2 * (int64_t)rec.idx + comp] * scale;
    const T kx = lane_on ? horner_cell(horner, nc, dx < w ? dx : 0, rec.z0) : (T)0;
    const int gx = wrap1(o0 + (int)(rec.loc & 1023) + dx, g.nf[0]);
    const int b1 = o1 + (int)((rec.loc >> 10) & 1023);
    const int b2 = o2 + (int)((rec.loc >> 20) & 1023);
    const T vx = cv * kx;
    for (int r0 = 0; r0 < nrows; r0 += rows_per_pass) {
      const int r = r0 + rsub;
      const bool on = lane_on && r < nrows;
      const int rc = r < nrows ? r : 0;
      const int dz = RANK > 2 ? rc / w : 0;
      const int dy = RANK > 2 ? rc - dz * w : rc;
      T v = vx;
      if (RANK > 1) v *= horner_cell(horner, nc, dy, rec.z1);
      if (RANK > 2) v *= horner_cell(horner, nc, dz, rec.z2);
      const int g1 = RANK > 1 ? wrap1(b1 + dy, g.nf[1]) : 0;
      const int g2 = RANK > 2 ? wrap1(b2 + dz, g.nf[2]) : 0;
      if (on) glb_add(&out[2 * (gx + (int64_t)g.nf[0] * (g1 + (int64_t)g.nf[1] * g2)) + comp], v);
    }
  }
}

// --------------------------------- spread: wavefront-per-point path (2D, w=8)

// Tile 32x32 fine cells (+7 one-sided halo) held as two PLANAR fp64 planes
// with a row stride of 40 elements. A wavefront handles one point per pass:
// lane (dy, dx) = (lane >> 3, lane & 7) owns one of the 8x8 stencil cells, so a
// pass is exactly two ds_add_f64 wave-instructions (re, im; fp64 because
// ds_add_f32 is ~22x slower on gfx950, see the generic kernel). With the row
// stride = 8 (mod 32) elements, i.e. 16 (mod 64) words, the 32 lanes of each
// half-wave (4 stencil rows x 8 cells x 2 words) cover all 64 banks exactly
// once: conflict free by construction, for every point position. Kernel values are produced 64 points at a time (one point
// per lane, Horner in registers) and handed to the per-point passes through
// a small LDS staging area read with broadcast loads.
constexpr int kWT = 32;              // tile edge
constexpr int kWW = 8;               // kernel width
constexpr int kWL = kWT + kWW - 1;   // 39 rows/cols used
constexpr int kWS = 40;              // row stride in words: 8 mod 32
constexpr int kWPlane = kWS * kWL;   // 1560 words per plane

template <int NW, int CH, bool FUSED = false>
__global__ __launch_bounds__(NW * 64) void spread_2d_w8_wave_kernel(
    Geom g, SortedPoints<float> sp, const float* __restrict__ horner, const float* __restrict__ c,
    float* __restrict__ fw, int64_t c_stride, int64_t fw_stride, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* plane_re = reinterpret_cast<double*>(smem_raw);
  double* plane_im = plane_re + kWPlane;
  float* stage_all = reinterpret_cast<float*>(plane_im + kWPlane);
  int tb, p0, p1, slot;
  if (!locate_subproblem(g, sp.tile_start, sp.sub_start, blockIdx.x, &tb, &p0, &p1, &slot)) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  for (int i = tid; i < 2 * kWPlane; i += NW * 64) plane_re[i] = 0.0;
  __syncthreads();

  float* kxs = stage_all + wave * (CH * kWW * 3);   // [CH][8]
  float2* kyc = reinterpret_cast<float2*>(kxs + CH * kWW);  // [CH][8] (ky*re, ky*im)
  const int dx = lane & 7, dy = lane >> 3;
  const int cell = dy * kWS + dx;
  const float* cc = c + 2 * (int64_t)slot * c_stride;

  // coefficients of the piecewise polynomial: uniform loads, kept in SGPRs/VGPRs
  for (int base = p0 + wave * CH; base < p1; base += NW * CH) {
    const int j = base + lane;
    const bool valid = lane < CH && j < p1;
    int off = 0;
    float kx[kWW], kyr[kWW], kyi[kWW];
    if (valid) {
      uint32_t loc;
      float zx, zy, re, im;
      if constexpr (FUSED) {   // the record carries the strength (FusedRec): no gather
        const FusedRec fr = reinterpret_cast<const FusedRec*>(sp.rec)[j];
        loc = fused_loc(fr.px, fr.py);
        zx = fused_z(fr.px);
        zy = fused_z(fr.py);
        re = fr.re * scale;
        im = fr.im * scale;
      } else {
        const Rec<float> rec = sp.rec[j];   // one 16-byte load: loc, zx, zy, idx
        loc = rec.loc;
        zx = rec.z0;
        zy = rec.z1;
        const float2 cv = reinterpret_cast<const float2*>(cc)[rec.idx];
        re = cv.x * scale;
        im = cv.y * scale;
      }
      off = ((loc >> 10) & 1023) * kWS + (loc & 1023);
#pragma unroll
      for (int q = 0; q < kWW; ++q) {
        float ax = horner[(kWaveCoef - 1) * kMaxW + q];
        float ay = ax;
#pragma unroll
        for (int k = kWaveCoef - 2; k >= 0; --k) {
          const float t = horner[k * kMaxW + q];
          ax = fmaf(ax, zx, t);
          ay = fmaf(ay, zy, t);
        }
        kx[q] = ax;
        kyr[q] = ay * re;
        kyi[q] = ay * im;
      }
    } else {
#pragma unroll
      for (int q = 0; q < kWW; ++q) { kx[q] = 0.f; kyr[q] = 0.f; kyi[q] = 0.f; }
    }
    // stage: each lane writes its point's 8 + 16 values
    if (lane < CH) {
      float4* d4 = reinterpret_cast<float4*>(kxs + lane * kWW);
      d4[0] = make_float4(kx[0], kx[1], kx[2], kx[3]);
      d4[1] = make_float4(kx[4], kx[5], kx[6], kx[7]);
      float4* e4 = reinterpret_cast<float4*>(kyc + lane * kWW);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        e4[q] = make_float4(kyr[2 * q], kyi[2 * q], kyr[2 * q + 1], kyi[2 * q + 1]);
    }
    // (same wave reads what it wrote: LDS ops of one wave are processed in order)
    int npts = p1 - base;
    if (npts > CH) npts = CH;
    const int nround = (npts + 3) & ~3;   // padded lanes hold zeros and off = 0
    for (int q = 0; q < nround; q += 4) {
      float a[4];
      float2 b[4];
      int o[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = kxs[(q + u) * kWW + dx];
        b[u] = kyc[(q + u) * kWW + dy];
        o[u] = __builtin_amdgcn_readlane(off, q + u) + cell;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        lds_add(&plane_re[o[u]], (double)(a[u] * b[u].x));
        lds_add(&plane_im[o[u]], (double)(a[u] * b[u].y));
      }
    }
  }
  __syncthreads();

  // Write-out: consecutive lanes take (re, im) of consecutive cells, so one
  // global_atomic_add_f32 wave-instruction covers 256 contiguous bytes of a
  // fine-grid row (the shape the memory-side atomic units want).
  const int t0 = tb % g.ntile[0];
  const int t1 = tb / g.ntile[0];
  const int o0 = t0 * kWT, o1 = t1 * kWT;
  float* out = fw + 2 * (int64_t)slot * fw_stride;
  for (int i = tid; i < 2 * kWL * kWL; i += NW * 64) {
    const int comp = i & 1;
    const int cellid = i >> 1;
    const int a0 = cellid % kWL, a1 = cellid / kWL;
    const float v = (float)(comp ? plane_im : plane_re)[a1 * kWS + a0];
    if (v != 0.f) {
      int g0 = o0 + a0; if (g0 >= g.nf[0]) g0 -= g.nf[0];
      int g1 = o1 + a1; if (g1 >= g.nf[1]) g1 -= g.nf[1];
      glb_add(&out[2 * (g0 + (int64_t)g.nf[0] * g1) + comp], v);
    }
  }
}

// Exclusive scan of cnt[NK] (NK = 1024 or 2048) in place by a workgroup of NT threads; the
// first NS = largest power of two <= NT threads do the work. wsum: >= NS/64 words of LDS
// scratch. Ends with a barrier.
template <int NT>
__device__ __forceinline__ void scan1024(uint32_t* cnt, uint32_t* wsum, int tid) { scan_counts<NT, 1024>(cnt, wsum, tid); }

// Second sort level, run once per set_points for dense 2-D point sets: every
// subproblem (<= 4096 points of one 32 x 32 tile) is counting-sorted by stencil
// start cell (1024 keys) with LDS integer atomics; records move from `in` to
// `out`. Points that share a start cell share the 8 x 8 patch address, which
// lets the spread kernel accumulate them in registers before one LDS atomic,
// and gives neighbouring interp threads neighbouring LDS addresses.
constexpr int kCellSortMaxSub = 4096;
constexpr int kCellSortThreads = 512;
template <typename T>
__global__ __launch_bounds__(kCellSortThreads) void cellsort2d_kernel(
    Geom g, const int32_t* __restrict__ tile_start, const int32_t* __restrict__ sub_start,
    const Rec<T>* __restrict__ in, Rec<T>* __restrict__ out) {
  constexpr int NT = kCellSortThreads;
  constexpr int IT = kCellSortMaxSub / NT;
  __shared__ uint32_t cnt[1024];
  __shared__ uint32_t wsum[16];
  int tb, p0, p1, slot;
  if (!locate_subproblem(g, tile_start, sub_start, blockIdx.x, &tb, &p0, &p1, &slot)) return;
  const int n = p1 - p0;
  const int tid = threadIdx.x;
  for (int i = tid; i < 1024; i += NT) cnt[i] = 0u;
  // loads are unconditional on clamped indices so that all of them are in flight together
  // (records as 16-byte vectors, every slot initialised: an array of Rec<T> that is assigned under a condition is
  // kept in scratch memory -- 48-172 bytes per lane, r04 verdict; `loc` is the first word of both record types)
  constexpr int NV = (int)(sizeof(Rec<T>) / 16);
  uint4 r[IT][NV];
  const uint4* inv = reinterpret_cast<const uint4*>(in);
  uint4* outv = reinterpret_cast<uint4*>(out);
#pragma unroll
  for (int u = 0; u < IT; ++u) {
    const int i = tid + u * NT;
    const size_t e = (size_t)(p0 + (i < n ? i : n - 1)) * NV;
#pragma unroll
    for (int k = 0; k < NV; ++k) r[u][k] = (u * NT < n) ? inv[e + k] : make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();
  uint32_t kr[IT];   // key | rank-in-cell << 10
#pragma unroll
  for (int u = 0; u < IT; ++u) {
    const int i = tid + u * NT;
    kr[u] = 0u;
    if (u * NT < n) {
      const uint32_t loc = r[u][0].x;
      const uint32_t key = (((loc >> 10) & 31u) << 5) | (loc & 31u);
      if (i < n) kr[u] = key | (atomicAdd(&cnt[key], 1u) << 10);
    }
  }
  __syncthreads();
  scan1024<NT>(cnt, wsum, tid);
#pragma unroll
  for (int u = 0; u < IT; ++u) {
    const int i = tid + u * NT;
    if (u * NT < n && i < n) {
      const size_t e = (size_t)(p0 + cnt[kr[u] & 1023u] + (kr[u] >> 10)) * NV;
#pragma unroll
      for (int k = 0; k < NV; ++k) outv[e + k] = r[u][k];
    }
  }
}

// The 3-D counterpart (tile 16 x 16 x 4|8: 2048 keys, x fastest). Its consumer is the
// thread-per-point interp kernel: neighbouring threads then read neighbouring (or the
// same) LDS cells, which removes the bank conflicts that bound its stencil loop.
template <typename T>
__global__ __launch_bounds__(kCellSortThreads) void cellsort3d_kernel(
    Geom g, const int32_t* __restrict__ tile_start, const int32_t* __restrict__ sub_start,
    const Rec<T>* __restrict__ in, Rec<T>* __restrict__ out) {
  constexpr int NT = kCellSortThreads;
  constexpr int IT = kCellSortMaxSub / NT;
  __shared__ uint32_t cnt[2048];
  __shared__ uint32_t wsum[16];
  int tb, p0, p1, slot;
  if (!locate_subproblem(g, tile_start, sub_start, blockIdx.x, &tb, &p0, &p1, &slot)) return;
  const int n = p1 - p0;
  const int tid = threadIdx.x;
  for (int i = tid; i < 2048; i += NT) cnt[i] = 0u;
  constexpr int NV = (int)(sizeof(Rec<T>) / 16);   // (records as 16-byte vectors: see cellsort2d_kernel)
  uint4 r[IT][NV];
  const uint4* inv = reinterpret_cast<const uint4*>(in);
  uint4* outv = reinterpret_cast<uint4*>(out);
#pragma unroll
  for (int u = 0; u < IT; ++u) {
    const int i = tid + u * NT;
    const size_t e = (size_t)(p0 + (i < n ? i : n - 1)) * NV;
#pragma unroll
    for (int k = 0; k < NV; ++k) r[u][k] = (u * NT < n) ? inv[e + k] : make_uint4(0u, 0u, 0u, 0u);
  }
  __syncthreads();
  uint32_t kr[IT];   // key | rank-in-cell << 11
#pragma unroll
  for (int u = 0; u < IT; ++u) {
    const int i = tid + u * NT;
    kr[u] = 0u;
    if (u * NT < n) {
      uint32_t loc;
      if constexpr (sizeof(T) == 4) {   // (packed float records: 4-bit tile-local starts in the top bits of three words)
        loc = (r[u][0].x >> 28) | ((r[u][0].y >> 28) << 10) | ((r[u][0].z >> 28) << 20);
      } else {
        loc = r[u][0].x;
      }
      const uint32_t key = (loc & 15u) | (((loc >> 10) & 15u) << 4) | (((loc >> 20) & 7u) << 8);
      if (i < n) kr[u] = key | (atomicAdd(&cnt[key], 1u) << 11);
    }
  }
  __syncthreads();
  scan_counts<NT, 2048>(cnt, wsum, tid);
#pragma unroll
  for (int u = 0; u < IT; ++u) {
    const int i = tid + u * NT;
    if (u * NT < n && i < n) {
      const size_t e = (size_t)(p0 + cnt[kr[u] & 2047u] + (kr[u] >> 11)) * NV;
#pragma unroll
      for (int k = 0; k < NV; ++k) outv[e + k] = r[u][k];
    }
  }
}

// Cell-grouped variant of the kernel above for dense point sets (>~ 0.5 points per
// fine cell). The subproblem's points are ordered by stencil start cell, either
// already in HBM (PRE: cellsort2d_kernel ran at set_points) or by the same
// counting sort done here in LDS. Consecutive passes of a wave that share a
// start cell then accumulate their 8x8 products in registers and issue ONE pair
// of ds_add_f64 per group instead of one per point: at config 2's density (2.4
// points per cell) that removes ~60 % of the LDS atomics that bound the
// ungrouped kernel.
constexpr int kGroupMaxSub = 4096;
// Phase timestamps of the grouped kernel's workgroups (experiment build -DNUFFT_HIP_PHASE_LOG, tools/phase_log_experiment.sh):
// thread 0 of every workgroup stores s_memtime at the phase boundaries; the product build compiles none of it.
#ifdef NUFFT_HIP_PHASE_LOG
constexpr int kPhaseSlots = 8, kPhaseLogWgs = 16384;
__device__ unsigned long long g_phase_log[kPhaseLogWgs * kPhaseSlots];
#define NUFFT_PHASE(k) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < kPhaseLogWgs) g_phase_log[blockIdx.x * kPhaseSlots + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define NUFFT_PHASE(k) do { } while (0)
#endif
constexpr double kGroupMinDensity = 0.5;   // points per fine cell
constexpr double kInterpSortMinDensity = 0.3;   // 3-D interp cell sort pays from here (r01: 0.075 loses, 0.75 and 1.8 win)
#ifndef NUFFT_GROUP_EXP   // (experiment builds, tools/group_loop_experiment.sh: pieces of the main loop left out -- wrong results, timing only;
#define NUFFT_GROUP_EXP 0 //  1 no LDS atomics, 2 no staging reads, 4 no kernel evaluation / staging writes)
#endif
#ifndef NUFFT_GROUP_STAGE   // (experiment builds: tools/group_shape_experiment.sh)
#define NUFFT_GROUP_STAGE 32
#endif
#ifndef NUFFT_GROUP_NW
#define NUFFT_GROUP_NW 12
#endif
constexpr int kGroupStage = NUFFT_GROUP_STAGE; // points whose kernel values are in LDS at a time (per wave; 16 measured 12 % slower, r02)
template <typename T> constexpr int kGroupStageOf = sizeof(T) == 8 ? 16 : kGroupStage;   // double: half, same bytes
constexpr int kGroupBlk = 36;   // staging words per block of 4 points (8 x 4 + 4 pad)
template <int CH> constexpr int kGroupStageWave = 3 * (CH / 4) * kGroupBlk;   // words per wave (kx, ky re, ky im)
// FUSED: the records are FusedRec (strength inside, no index): no gather at all.
// (r03 measured this kernel on 64 x 64 tiles -- 82 KB of planes, 4096 start cells, one workgroup per CU -- so that the
// sort becomes ONE staged pass: scatter 161 -> 97 us, spread 283 -> 397 us at config 2, a net loss; EXPERIMENTS.md section 5.)
template <typename T, int W, int NW, int CH, bool PRE, bool FUSED = false>
__global__ __launch_bounds__(NW * 64) void spread_2d_w8_group_kernel(
    Geom g, SortedPoints<T> sp, const T* __restrict__ horner, const T* __restrict__ c,
    T* __restrict__ fw, int64_t c_stride, int64_t fw_stride, T scale) {
  typedef T v2 __attribute__((ext_vector_type(2)));
  typedef T v4 __attribute__((ext_vector_type(4)));
  using RecT = std::conditional_t<FUSED, FusedRec, Rec<T>>;
  const RecT* __restrict__ recs = reinterpret_cast<const RecT*>(sp.rec);
  constexpr int NT = NW * 64;
  constexpr int IT = (kGroupMaxSub + NT - 1) / NT;   // records per thread in the LDS sort
  constexpr int SC = CH < kGroupStageOf<T> ? CH : kGroupStageOf<T>;   // points staged through LDS at a time
  static_assert(NT <= 1024, "at most 16 waves");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* plane_re = reinterpret_cast<double*>(smem_raw);
  double* plane_im = plane_re + kWPlane;
  T* stage_all = reinterpret_cast<T*>(plane_im + kWPlane);
  uint32_t* cnt = reinterpret_cast<uint32_t*>(stage_all + NW * kGroupStageWave<SC>);   // [1024]
  uint16_t* perm = reinterpret_cast<uint16_t*>(cnt + 1024);                     // [4096]
  uint32_t* wsum = reinterpret_cast<uint32_t*>(perm + kGroupMaxSub);            // [16]
  int tb, p0, p1, slot;
  NUFFT_PHASE(0);
  if (!locate_subproblem(g, sp.tile_start, sp.sub_start, blockIdx.x, &tb, &p0, &p1, &slot)) return;
  NUFFT_PHASE(1);
  const int n = p1 - p0;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  for (int i = tid; i < 2 * kWPlane; i += NT) plane_re[i] = 0.0;
  if constexpr (!PRE) {
    for (int i = tid; i < 1024; i += NT) cnt[i] = 0u;
    __syncthreads();
    NUFFT_PHASE(2);

    // ---- LDS counting sort of the subproblem by start cell (plans whose records are
    // not already cell-ordered by cellsort2d_kernel)
    // (all loads are unconditional on clamped indices: a load under a divergent
    // branch makes the compiler wait for it before the next one is issued. Issuing them
    // before the zero-fill was measured, r03: the wait moves into the zero-fill phase, the
    // workgroup takes the same 66.7 k cycles.)
    uint32_t kr[IT];   // key | rank-in-cell << 10
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const int i = tid + u * NT;
      kr[u] = 0u;
      if (u * NT < n) {
        if constexpr (FUSED) {
          const uint2 pp = *reinterpret_cast<const uint2*>(&recs[p0 + (i < n ? i : n - 1)]);
          kr[u] = fused_loc(pp.x, pp.y);
        } else {
          kr[u] = recs[p0 + (i < n ? i : n - 1)].loc;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const int i = tid + u * NT;
      const uint32_t key = (((kr[u] >> 10) & 31u) << 5) | (kr[u] & 31u);
      if (i < n) kr[u] = key | (atomicAdd(&cnt[key], 1u) << 10);
    }
    __syncthreads();
    NUFFT_PHASE(3);
    scan1024<NT>(cnt, wsum, tid);
    NUFFT_PHASE(4);
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const int i = tid + u * NT;
      if (i < n) perm[cnt[kr[u] & 1023u] + (kr[u] >> 10)] = (uint16_t)i;
    }
  }
  __syncthreads();
  NUFFT_PHASE(5);

  // Staging holds kernel values of 4 consecutive points side by side, so one
  // ds_read_b128 fetches a lane's kx (or ky) for 4 passes (6 cycles instead of
  // 4 x 3.4, tools/ubench/lds_read_bench.hip). Block stride 36 words: the
  // lane-per-point b32 writes of one q then hit 64 distinct banks.
  T* kxs = stage_all + wave * kGroupStageWave<SC>;   // [CH/4][8][4] (+4 pad per block)
  T* kyr = kxs + kGroupStageWave<SC> / 3;            // ky * re(c)
  T* kyi = kyr + kGroupStageWave<SC> / 3;            // ky * im(c)
  const int dx = lane & 7, dy = lane >> 3;
  const int cell = (dy * kWS + dx) * (int)sizeof(double);
  const T* cc = c + 2 * (int64_t)slot * c_stride;

  // Two-deep software pipeline over the wave's chunks: the record gather of chunk
  // i+2 and the strength gather of chunk i+1 (which needs record i+1's index)
  // are in flight while chunk i is spread, so neither HBM latency is exposed.
  const v2* c2 = reinterpret_cast<const v2*>(cc);
  auto load_rec = [&](int b) {   // lanes past the end re-read the last point; masked below
    const int li = b + lane;
    const int lc = li < n ? li : n - 1;
    if constexpr (PRE) return recs[p0 + lc];
    else return recs[p0 + perm[lc]];
  };
  auto load_c = [&](const RecT& r) {
    if constexpr (FUSED) { v2 v = {(T)r.re, (T)r.im}; return v; }
    else return c2[r.idx];
  };
  // (Equal CONTIGUOUS shares per wave instead of round-robin chunks measured slower, r02:
  // 283 -> 295 us at config 2 -- every wave then pays a fourth, mostly empty, phase 1.)
  const int first = wave * CH;
  const int limit = n;
  constexpr int STEP = NW * CH;
  RecT r_cur = load_rec(first);
  RecT r_nxt = load_rec(first + STEP);
  v2 c_cur = load_c(r_cur);
  for (int base = first; base < limit; base += STEP) {
    const v2 c_nxt = load_c(r_nxt);
    const RecT r_nn = load_rec(base + 2 * STEP);
    uint32_t loc;
    T zx, zy;
    if constexpr (FUSED) {
      loc = fused_loc(r_cur.px, r_cur.py);
      zx = (T)fused_z(r_cur.px);
      zy = (T)fused_z(r_cur.py);
    } else {
      loc = r_cur.loc;
      zx = r_cur.z0;
      zy = r_cur.z1;
    }
    const bool valid = lane < CH && base + lane < limit;
    const T re = valid ? c_cur.x * scale : (T)0, im = valid ? c_cur.y * scale : (T)0;
    const uint32_t key = valid ? (loc & 0xfffffu) : 0xffffffffu;
    const int off = (((loc >> 10) & 1023) * kWS + (loc & 1023)) * (int)sizeof(double);
    T kx[kWW], ky[kWW];
    // The ES kernel is even, so cell W-1-q's polynomial is cell q's at -z: evaluate the
    // even and odd parts in z^2 once per pair (12 instead of 22 FMAs for two cells,
    // and at most 48 coefficients, which stay in SGPRs). x and y share coefficients:
    // each term is one v_pk_fma_f32 (two v_fma_f64 in double).
    const v2 zz = {zx, zy};
    const v2 z2 = zz * zz;
#pragma unroll
    for (int q = 0; q < kWW; ++q) { kx[q] = (T)0; ky[q] = (T)0; }   // cells >= W stay 0 (narrower kernels)
#pragma unroll
    for (int q = 0; q < (W + 1) / 2; ++q) {
      const T te = horner[(kWaveCoef - 2) * kMaxW + q], to = horner[(kWaveCoef - 1) * kMaxW + q];
      v2 e = {te, te}, o = {to, to};
#pragma unroll
      for (int m = kWaveCoef / 2 - 2; m >= 0; --m) {
        const T ce = horner[(2 * m) * kMaxW + q], co = horner[(2 * m + 1) * kMaxW + q];
        const v2 cce = {ce, ce}, cco = {co, co};
        e = __builtin_elementwise_fma(e, z2, cce);
        o = __builtin_elementwise_fma(o, z2, cco);
      }
      const v2 lo = __builtin_elementwise_fma(zz, o, e), hi = __builtin_elementwise_fma(-zz, o, e);
      kx[q] = lo.x; ky[q] = lo.y;
      if (W - 1 - q != q) { kx[W - 1 - q] = hi.x; ky[W - 1 - q] = hi.y; }
    }
    r_cur = r_nxt;
    c_cur = c_nxt;
    r_nxt = r_nn;
    int npts = limit - base;
    if (npts > CH) npts = CH;
    // a pass ends a group when the next point starts in a different cell (or the staged half ends)
    const uint32_t nxt = __shfl_down(key, 1);
    const unsigned long long tailm =
        __ballot(lane < npts && (lane == npts - 1 || (lane & (SC - 1)) == SC - 1 || nxt != key));
    // Kernel values of the chunk stay in registers; they go through LDS SC points
    // at a time so that the staging area stays small (LDS bytes decide how many
    // workgroups share a CU).
#pragma unroll 1
    for (int h = 0; h < CH; h += SC) {
      if (h >= npts) break;
      if (!(NUFFT_GROUP_EXP & 4) && lane >= h && lane < h + SC) {
        const int sl = lane - h;
        const int sb = (sl >> 2) * kGroupBlk + (sl & 3);
#pragma unroll
        for (int q = 0; q < kWW; ++q) {
          kxs[sb + 4 * q] = kx[q];
          kyr[sb + 4 * q] = ky[q] * re;
          kyi[sb + 4 * q] = ky[q] * im;
        }
      }
      int nh = npts - h;
      if (nh > SC) nh = SC;
      const unsigned tails = (unsigned)(tailm >> h);
      const int nround = (nh + 3) & ~3;   // padded lanes hold zeros and are never tails
      T ar = (T)0, ai = (T)0;
      for (int q = 0; q < nround; q += 4) {
#if NUFFT_GROUP_EXP & 2
        const v4 ax4 = {(T)q, (T)1, (T)2, (T)3}, br4 = {(T)dx, (T)dy, (T)1, (T)2}, bi4 = {(T)dy, (T)dx, (T)3, (T)4};
#else
        const v4 ax4 = *reinterpret_cast<const v4*>(kxs + (q >> 2) * kGroupBlk + 4 * dx);
        const v4 br4 = *reinterpret_cast<const v4*>(kyr + (q >> 2) * kGroupBlk + 4 * dy);
        const v4 bi4 = *reinterpret_cast<const v4*>(kyi + (q >> 2) * kGroupBlk + 4 * dy);
#endif
        const T a[4] = {ax4.x, ax4.y, ax4.z, ax4.w};
        const T br[4] = {br4.x, br4.y, br4.z, br4.w};
        const T bi[4] = {bi4.x, bi4.y, bi4.z, bi4.w};
        const unsigned t4 = (tails >> q) & 15u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          // (broadcasting the strength with v_readlane instead of staging ky * re and ky * im
          // separately saves a ds_read_b128 per four points and measured 3 % slower, r02: VALU)
          ar = fma(a[u], br[u], ar);
          ai = fma(a[u], bi[u], ai);
          if (t4 & (1u << u)) {
            const int o = __builtin_amdgcn_readlane(off, h + q + u) + cell;   // byte offset into the re plane
#if NUFFT_GROUP_EXP & 1
            if (ar == (T)123.456) plane_re[o & 255] = 1.0;
#else
            lds_add(reinterpret_cast<double*>(smem_raw + o), (double)ar);
            lds_add(reinterpret_cast<double*>(smem_raw + o) + kWPlane, (double)ai);
#endif
            ar = (T)0;
            ai = (T)0;
          }
        }
      }
    }
  }
  __syncthreads();
  NUFFT_PHASE(6);

  const int t0 = tb % g.ntile[0];
  const int t1 = tb / g.ntile[0];
  const int o0 = t0 * kWT, o1 = t1 * kWT;
  T* out = fw + 2 * (int64_t)slot * fw_stride;
  for (int i = tid; i < 2 * kWL * kWL; i += NT) {
    const int comp = i & 1;
    const int cellid = i >> 1;
    const int a0 = cellid % kWL, a1 = cellid / kWL;
    const T v = (T)(comp ? plane_im : plane_re)[a1 * kWS + a0];
    if (v != (T)0) {
      int g0 = o0 + a0; if (g0 >= g.nf[0]) g0 -= g.nf[0];
      int g1 = o1 + a1; if (g1 >= g.nf[1]) g1 -= g.nf[1];
      glb_add(&out[2 * (g0 + (int64_t)g.nf[0] * g1) + comp], v);
    }
  }
  NUFFT_PHASE(7);
}


// --------------------- spread: 2-D wavefront path with compile-time width

// The 2-D kernel for every width <= 8 and both precisions, in the branch-free
// form of the 3-D kernel below: tile 32 x 32, row stride 40, lanes outside the
// W x W patch add 0 at their natural 8 x 8 patch address (rows beyond the tile
// fall into the next component plane or the 256-element pad behind the planes).
template <typename T, int W, int NW, int CH, bool FUSED = false>
__global__ __launch_bounds__(NW * 64) void spread_wave2_kernel(
    Geom g, SortedPoints<T> sp, const T* __restrict__ horner, const T* __restrict__ c,
    T* __restrict__ fw, int64_t c_stride, int64_t fw_stride, T scale) {
  using T2 = typename Pair<T>::type;
  constexpr int LS = 40, L0 = 32 + W - 1, L1 = 32 + W - 1;
  constexpr int plane = LS * L1;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* plane_re = reinterpret_cast<double*>(smem_raw);
  double* plane_im = plane_re + plane;
  T* stage_all = reinterpret_cast<T*>(plane_im + plane + 256);
  int tb, p0, p1, slot;
  if (!locate_subproblem(g, sp.tile_start, sp.sub_start, blockIdx.x, &tb, &p0, &p1, &slot)) return;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  for (int i = tid; i < 2 * plane + 256; i += NW * 64) plane_re[i] = 0.0;
  __syncthreads();

  const int nc = g.ncoef;
  T* kxs = stage_all + wave * (CH * 24);                     // [CH][8]
  T2* kyc = reinterpret_cast<T2*>(kxs + CH * 8);             // [CH][8] (ky*re, ky*im)
  const int dx = lane & 7, dy = lane >> 3;
  const bool active = dx < W && dy < W;
  const int cell = dy * LS + dx;
  const T2* cc = reinterpret_cast<const T2*>(c) + (int64_t)slot * c_stride;

  for (int base = p0 + wave * CH; base < p1; base += NW * CH) {
    const int j = base + lane;
    int off = 0;
    if (lane < CH) {
      T kx[8], ky[8], kdummy[8];
      T re = (T)0, im = (T)0;
#pragma unroll
      for (int q = 0; q < 8; ++q) { kx[q] = (T)0; ky[q] = (T)0; }
      if (j < p1) {
        PointView<T> rec;
        if constexpr (FUSED) {
          const FusedRec fr = reinterpret_cast<const FusedRec*>(sp.rec)[j];
          rec.loc = fused_loc(fr.px, fr.py);
          rec.z0 = (T)fused_z(fr.px);
          rec.z1 = (T)fused_z(fr.py);
          re = (T)fr.re * scale;
          im = (T)fr.im * scale;
        } else {
          rec = unpack_rec<T, 2>(sp.rec[j]);
          const T2 cv = cc[rec.idx];
          re = cv.x * scale;
          im = cv.y * scale;
        }
        off = (int)(rec.loc & 1023) + (int)((rec.loc >> 10) & 1023) * LS;
        horner8<T, 2>(horner, nc, rec.z0, rec.z1, (T)0, kx, ky, kdummy);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        kxs[lane * 8 + q] = kx[q];
        T2 v; v.x = ky[q] * re; v.y = ky[q] * im;
        kyc[lane * 8 + q] = v;
      }
    }
    int npts = p1 - base;
    if (npts > CH) npts = CH;
    const int nround = (npts + 3) & ~3;   // padded slots hold zeros and off = 0
    for (int q = 0; q < nround; q += 4) {
      T a[4]; T2 b[4]; int o[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = kxs[(q + u) * 8 + dx];
        b[u] = kyc[(q + u) * 8 + dy];
        o[u] = __builtin_amdgcn_readlane(off, q + u) + cell;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const T wa = active ? a[u] : (T)0;   // (all 8 staged values are written, but keep the patch exact)
        lds_add(&plane_re[o[u]], (double)(wa * b[u].x));
        lds_add(&plane_im[o[u]], (double)(wa * b[u].y));
      }
    }
  }
  __syncthreads();
  Geom gl = g;   // tile_to_grid reads the extents from the geometry
  (void)L0;
  tile_to_grid<T, 2>(gl, plane_re, plane_im, LS, plane, tb, fw + 2 * (int64_t)slot * fw_stride, wave, NW, lane);
}

// ------------------------- spread: 3-D wavefront path with compile-time width

// The 3-D counterpart of spread_wave2_kernel, with the kernel width W and the
// tile depth TZ as template parameters (tile 16 x 16 x TZ, row stride 24): every
// LDS offset of the per-point body is an immediate and the body is straight-line
// code -- no exec-masked branch (lanes outside the W x W patch add 0 at their
// natural patch address), so the only LDS wait in the loop is a COUNTED one on the
// prefetched kx/ky reads and the atomics of consecutive points stream back to
// back. History (r01, config 4): run-time width, LDS reads for the z factors,
// lgkmcnt(0) after every point: 35 ms, 45 % LDS-array activity; W-templated
// with v_readlane broadcasts: 24 ms; this form: see DESIGN.md.
//
// FX = false: two fp64 planes, 2 W ds_add_f64 per point.
// FX = true (float, W <= 6, i.e. tol >= ~1e-4): ONE 64-bit integer per cell
// holding (re, im) as two signed 32-bit fixed-point fields, W ds_add_u64 per
// point. X = re_i 2^32 + im_i is added exactly (the sign extension of im_i is
// folded into the upper half), so the fields cannot interfere while each sum
// fits in 32 bits. The scale is per subproblem: 2^31 / sum_j max(|re c_j|,
// |im c_j|), the true worst-case bound of any cell (kernel values <= 1), from
// a first pass over the subproblem's strengths -- one huge strength only
// coarsens its own subproblem. Quantisation <= 0.5 LSB per contribution; worst
// case 2^-32 n w^1.5 ||c_sub|| (n <= 4096: 1.4e-5), typically ~1e-6 relative.
// COMP (fp64 planes only): 0 = both components in one launch (two planes); 1 / 2 = only
// the real / imaginary part (ONE plane, so two workgroups fit a CU; the host launches both).
template <int W> constexpr int kWave3Pad = (551 - 24 * (15 + W) + 1) > 64 ? ((551 - 24 * (15 + W) + 1 + 7) & ~7) : 64;
// sub: the subproblem (launch slot) this call works on -- blockIdx.x, or an entry of the fallback list (see the kernel)
constexpr int kCrowdJoinFrom = 64, kCrowdJoinMax = 8;   // (spread_wave3_body: subproblems of a crowded tile joined per workgroup)
template <typename T, int W, int TZ, int NW, int CH, bool FX, int COMP>
__device__ __forceinline__ void spread_wave3_body(
    const Geom& g, const SortedPoints<T>& sp, const T* __restrict__ horner, const T* __restrict__ c,
    T* __restrict__ fw, int64_t c_stride, int64_t fw_stride, T scale, int sub, bool listed) {
  using T2 = typename Pair<T>::type;
  constexpr int LS = 24, L0 = 16 + W - 1, L1 = 16 + W - 1, L2 = TZ + W - 1;
  constexpr int PS = LS * L1;
  constexpr int plane = PS * L2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* plane_re = reinterpret_cast<double*>(smem_raw);             // FX: the packed plane
  constexpr int NPL = (FX || COMP != 0) ? 1 : 2;
  double* plane_im = plane_re + (NPL == 1 ? 0 : plane);
  // spill room behind the planes: lanes outside the W x W patch add 0 at their natural 8 x 8
  // patch address, up to row 15 + 7 and column 15 + 7 of the last z-plane, i.e. up to
  // 22 * 24 + 22 - 24 * L1 elements past the end (47 at W = 6, 143 at W = 2)
  constexpr int PAD = kWave3Pad<W>;
  double* pad = plane_re + NPL * plane;
  T* stage_all = reinterpret_cast<T*>(pad + PAD);
  constexpr int SW = W <= 6 ? 6 : 8;                                  // staging row length
  float* red = reinterpret_cast<float*>(stage_all + NW * CH * 2 * SW);   // [NW] (FX bound reduction)
  // (behind a fused 3-D sort the records are 32-byte FusedRec3: the 16-byte record comes first)
  const int rstride = g.fused ? (int)sizeof(FusedRec3) : (int)sizeof(Rec<T>);
  int tb, p0, p1, slot, nsub, chunk, tile_end;
  if (!locate_subproblem(g, sp.tile_start, sp.sub_start, sub, &tb, &p0, &p1, &slot, &nsub, &chunk, &tile_end)) return;
  // Fixed-point plans: crowded tiles go to the fp64-plane kernels. The quantisation noise of the
  // packed fields is the same in every cell of a subproblem's tile, whatever the cell's kernel
  // weight, and every further subproblem of the tile adds its share: with 600000 coincident points
  // (1172 subproblems in one tile) the transform missed tol = 1e-5 by 9x (8.8e-5 against 1.4e-6 with
  // fp64 planes; r02 soak, seed 45), while up to ~16 subproblems per tile it stays within a third of tol.
  // (subproblems that reach this function through the fallback list were chosen by set_points: bound3_kernel /
  // crowded_list_kernel)
  if (!listed && g.fixed_point && (FX ? nsub > g.fx_max_subs : nsub <= g.fx_max_subs)) return;
  if constexpr (!FX && sizeof(T) == 4) {
    // Very crowded tiles, float fine grid: every subproblem adds its partial sums to the same cells with float
    // atomics, ~3.3e-8 sqrt(subproblems) of rounding (r04 soak: 1.5e6 points in one tile = 586 subproblems: 1.3e-6
    // at tol 1e-6). The planes are fp64 and nothing here is sized by the point count: above kCrowdJoinFrom
    // subproblems, up to kCrowdJoinMax consecutive ones are accumulated by the workgroup of the first and written
    // out once (the others exit) -- fewer, longer workgroups, on point sets that have one hot tile anyway.
    if (nsub > kCrowdJoinFrom) {
      int join = (nsub + kCrowdJoinFrom - 1) / kCrowdJoinFrom;
      if (join > kCrowdJoinMax) join = kCrowdJoinMax;
      if (chunk % join) return;
      const long long end = (long long)p0 + (long long)join * (p1 - p0);   // (an even split: p1 - p0 is the chunk size unless this is the last one)
      p1 = end < (long long)tile_end ? (int)end : tile_end;
    }
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  for (int i = tid; i < NPL * plane + PAD; i += NW * 64) plane_re[i] = 0.0;
  const T2* cc = reinterpret_cast<const T2*>(c) + (int64_t)slot * c_stride;
  const int npt = p1 - p0;

  T pre = scale;     // multiplies the strengths (FX: also converts to LSB units)
  T lsb = (T)1;
  if (FX) {
    float part = 0.f, big = 0.f;
    for (int j = p0 + tid; j < p1; j += NW * 64) {
      const T2 cv = cc[unpack_rec<T, 3>(rec_at(sp.rec, j, rstride)).idx];
      const float m = fmaxf(fabsf((float)cv.x), fabsf((float)cv.y));
      part += m;
      big = fmaxf(big, m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      part += __shfl_down(part, o);
      big = fmaxf(big, __shfl_down(big, o));
    }
    if (lane == 0) { red[wave] = part; red[NW + wave] = big; }
    __syncthreads();
    float bound = 0.f, top = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) { bound += red[k]; top = fmaxf(top, red[NW + k]); }
    // every cell sum is bounded by sum_j |c_j| prod_d max|P_d|; the fitted polynomials overshoot 1
    // slightly (w = 4: 1.00001 per dimension), which g.fx_headroom (>= 1, from the host fit) covers
    const float amp = fabsf((float)scale) * g.fx_headroom;
    const float room = 2147483000.f - (float)npt;   // 2^31 minus the rounding of every contribution
    // The step obeys the sum rule (no cell overflows 32 bits) and, for strengths of similar size (largest <= 8 x the
    // mean), the top rule (every contribution below 2^22 steps: the exact range of the FMA conversion in the loop).
    // With one dominant strength it follows the mean instead, and the strengths above 2^22 steps are added behind
    // the loop with the exact conversion (nufft_dense3.hip has the same rule and the reasoning).
    const float s_sum = bound * amp / room;
    const bool skewed = top * (float)npt > 8.f * bound;
    const float step = fmaxf(s_sum, (skewed ? 2.f * bound / (float)npt : top) * amp * (1.f / 4194000.f));
    pre = step > 0.f ? (T)((float)scale / step) : (T)0;
    lsb = (T)step;
  }
  __syncthreads();

  const int nc = g.ncoef;
  T* kxs = stage_all + wave * (CH * 2 * SW);   // [CH][SW]
  T* kys = kxs + CH * SW;                      // [CH][SW]
  const int dx = lane & 7, dy = lane >> 3;
  const bool active = dx < W && dy < W;
  const int cell = dy * LS + dx;
  const int share = (npt + NW - 1) / NW;
  const int wbeg = p0 + wave * share;
  const int wend = (wbeg + share < p1) ? wbeg + share : p1;

  for (int base = wbeg; base < wend; base += CH) {
    const int j = base + lane;
    int off = 0;
    T kz[W];
    T cre = (T)0, cim = (T)0, bre = (T)0, bim = (T)0;
#pragma unroll
    for (int q = 0; q < W; ++q) kz[q] = (T)0;
    if (lane < CH) {
      T kx[W], ky[W];
#pragma unroll
      for (int q = 0; q < W; ++q) { kx[q] = (T)0; ky[q] = (T)0; }
      if (j < wend) {
        const PointView<T> rec = unpack_rec<T, 3>(rec_at(sp.rec, j, rstride));
        const T2 cv = cc[rec.idx];
        cre = cv.x * pre;
        cim = cv.y * pre;
        if (FX) {   // too large for the FMA conversion (see the prelude): waits for the exact pass behind the loop
          if (fmaxf(fabsf((float)cre), fabsf((float)cim)) * g.fx_headroom > 4194000.f) { bre = cre; bim = cim; cre = (T)0; cim = (T)0; }
        }
        off = (int)(rec.loc & 1023) + (int)((rec.loc >> 10) & 1023) * LS + (int)((rec.loc >> 20) & 1023) * PS;
        T h0[8], h1[8], h2[8];
        horner8<T, 3>(horner, nc, rec.z0, rec.z1, rec.z2, h0, h1, h2);
#pragma unroll
        for (int q = 0; q < W; ++q) { kx[q] = h0[q]; ky[q] = h1[q]; kz[q] = h2[q]; }
      }
#pragma unroll
      for (int q = 0; q < W; ++q) {
        kxs[lane * SW + q] = kx[q];
        kys[lane * SW + q] = ky[q];
      }
    }
    int npts = wend - base;
    if (npts > CH) npts = CH;
    // raw staging values of the NEXT point are requested before this point's atomics
    T kx_n = kxs[dx], ky_n = kys[dy];
    for (int q = 0; q < npts; ++q) {
      // lanes outside the patch read unwritten staging slots: force their weight to 0
      const T a = active ? kx_n * ky_n : (T)0;
      const int qn = (q + 1 < npts) ? q + 1 : q;
      kx_n = kxs[qn * SW + dx];   // lanes outside the patch read a neighbour's slot; masked above
      ky_n = kys[qn * SW + dy];
      const int o = __builtin_amdgcn_readlane(off, q) + cell;
      const T ar = a * bcast_lane(cre, q);
      const T ai = a * bcast_lane(cim, q);
      // Lanes outside the W x W patch add 0 at their natural 8 x 8 patch address
      // (keeps the body branch-free, the offsets immediate and the bank pattern
      // conflict free). Those addresses stay inside LDS: columns < 24 = LS, and
      // rows >= L1 fall into the first rows of the next z-plane / the next
      // component plane / the 64-element pad behind the planes.
      double* pr = plane_re + o;
      double* pi = plane_im + o;
#pragma unroll
      for (int dz = 0; dz < W; ++dz) {
        const T kzq = bcast_lane(kz[dz], q);
        if (FX) {
          // float -> packed fixed point in TWO instructions: v_pk_fma_f32 onto the magic number
          // 1.5 * 2^23 leaves round-to-nearest(product) in the low mantissa bits of either half,
          // i.e. the register pair read as one 64-bit integer is (B + n_re) 2^32 + (B + n_im) with
          // B = 0x4B400000 and B + n_im > 0; subtracting the constant B (2^32 + 1) (one v_lshl_add_u64)
          // leaves n_re 2^32 + n_im, the sign extension of the low field folded into the high one.
          // (r02: v_pk_mul + 2 v_cvt_rpi + shift + add + pack = 6 instructions per atomic; the loop
          // was VALU-issue bound at 53 instructions per point.)
          typedef float v2f __attribute__((ext_vector_type(2)));
          const v2f fx = __builtin_elementwise_fma((v2f){(float)ai, (float)ar}, (v2f){(float)kzq, (float)kzq},
                                                   (v2f){12582912.f, 12582912.f});
          const unsigned long long x = __builtin_bit_cast(unsigned long long, fx) - 0x4B4000004B400000ull;
          atomicAdd(reinterpret_cast<unsigned long long*>(pr) + dz * PS, x);
        } else {
          if (COMP != 2) lds_add(pr + dz * PS, (double)(ar * kzq));
          if (COMP != 1) lds_add(pi + dz * PS, (double)(ai * kzq));
        }
      }
    }
    if constexpr (FX) {
      // strengths above 2^22 steps (only where one strength dominates its subproblem), one at a time with the exact
      // conversion. Kept out of the loop above: a per-point repeat count around its body cost 55 % at w = 7.
      unsigned long long pend = __ballot(bre != (T)0 || bim != (T)0);
      while (pend) {
        const int src = __ffsll((long long)pend) - 1;
        pend &= pend - 1;
        const T a = active ? kxs[src * SW + dx] * kys[src * SW + dy] : (T)0;
        const float ar = (float)(a * bcast_lane(bre, src)), ai = (float)(a * bcast_lane(bim, src));
        unsigned long long* pr = reinterpret_cast<unsigned long long*>(plane_re + __builtin_amdgcn_readlane(off, src) + cell);
#pragma unroll
        for (int dz = 0; dz < W; ++dz) {
          const float kzq = (float)bcast_lane(kz[dz], src);
          const int ir = __float2int_rn(ar * kzq), ii = __float2int_rn(ai * kzq);
          atomicAdd(pr + dz * PS, ((unsigned long long)(unsigned)(ir + (ii >> 31)) << 32) | (unsigned)ii);
        }
      }
    }
  }
  __syncthreads();

  // write-out: (FX: unpack, scale back,) add to the periodic fine grid
  int t0, t1, t2;
  tile_coords(g, tb, &t0, &t1, &t2);
  const int o0 = t0 * 16, o1 = t1 * 16, o2 = t2 * TZ;
  T* out = fw + 2 * (int64_t)slot * fw_stride;
  for (RowWalk r(wave, L1); r.a2 < L2; r.advance(NW, L1)) {
    const int g1 = wrap1(o1 + r.a1, g.nf[1]);
    const int g2 = wrap1(o2 + r.a2, g.nf[2]);
    const int64_t rowbase = (int64_t)g.nf[0] * (g1 + (int64_t)g.nf[1] * g2);
    const int lrow = r.a2 * PS + r.a1 * LS;
    for (int e = lane; e < 2 * L0; e += 64) {
      const int a0 = e >> 1, comp = e & 1;
      T v;
      if (FX) {
        const long long t = (long long)reinterpret_cast<const unsigned long long*>(plane_re)[lrow + a0];
        const int im_sum = (int)(unsigned)(t & 0xffffffffll);
        const int re_sum = (int)((t - (long long)im_sum) >> 32);
        v = (T)(comp ? im_sum : re_sum) * lsb;
      } else {
        if (COMP != 0 && comp != COMP - 1) continue;
        v = (T)(comp ? plane_im : plane_re)[lrow + a0];
      }
      if (v != (T)0) glb_add(&out[2 * (rowbase + wrap1(o0 + a0, g.nf[0])) + comp], v);
    }
  }
}

// The kernel: one subproblem per workgroup (blockIdx.x), or -- the fp64-plane launches BEHIND a fixed-point spreader,
// !FX on a fixed-point plan -- a small persistent grid walking the list of subproblems that set_points left to the
// fp64 planes (crowded tiles, bounds above the limit). r04: as full-grid launches whose workgroups exit on one scalar
// load these two launches cost ~75 us each at 85 000 workgroups (GRBM_GUI_ACTIVE), 3 % of the 3-D spread stage, for
// nothing in the common case; an empty list now costs a few microseconds.
template <typename T, int W, int TZ, int NW, int CH, bool FX, int COMP = 0>
__global__ __launch_bounds__(NW * 64) void spread_wave3_kernel(
    Geom g, SortedPoints<T> sp, const T* __restrict__ horner, const T* __restrict__ c,
    T* __restrict__ fw, int64_t c_stride, int64_t fw_stride, T scale) {
  if constexpr (!FX) {
    if (g.fixed_point && sp.fb_list) {
      const int n = sp.fb_list[0];   // (entries follow the count)
      for (int it = blockIdx.x; it < n; it += gridDim.x) {
        spread_wave3_body<T, W, TZ, NW, CH, FX, COMP>(g, sp, horner, c, fw, c_stride, fw_stride, scale, sp.fb_list[1 + it], true);
        __syncthreads();   // (the next subproblem zeroes the planes this one's write-out reads)
      }
      return;
    }
  }
  spread_wave3_body<T, W, TZ, NW, CH, FX, COMP>(g, sp, horner, c, fw, c_stride, fw_stride, scale, (int)blockIdx.x, false);
}

// ------------------------- spread: 3-D fp64 planes over STACKS of tiles (r06; double precision)
//
// A subproblem of spread_wave3_kernel writes its tile + halo to the fine grid: (16 + W - 1)^2 (TZ + W - 1) cells for a
// 16 x 16 x TZ tile -- 5.7 x the tile at W = 8 on the depth-4 tiles of the double-precision plans (an LDS cell is 16
// bytes there), as global fp64 atomics. Below ~1 point per fine cell that write-out, the zero-fill of 97 KB of
// planes and the launch of one workgroup per ~76 points (256^3 modes, M = 1e7) ARE the double-precision 3-D
// spreader: 10.1 ms against 2.4 ms for the float transform (profiles/r06_c128_before.txt). The stacks that r05 cut for
// the float fixed-point kernels (nufft_dense3.hip, stack_plan_kernel: runs of tiles of one (x, y) column,
// consecutive in z) serve here unchanged: one workgroup walks a stack, after tile t its TZ finished planes are
// written out, the W - 1 halo planes move down by TZ to be the first planes of tile t + 1 (a chain of up to
// ceil((TZ + W - 1) / TZ) planes per (y, x) column, moved by the one lane that also writes the column out) and the
// freed planes are zeroed. Cells written per tile: 23 x 23 x (4 + 7 / nz) instead of 23 x 23 x 11.
// Pieces (tiles above max_sub points, cut as locate_subproblem would) are stacks of one tile and a point range.
template <typename T, int W, int TZ, int NW, int CH>
__global__ __launch_bounds__(NW * 64) void spread_wave3_stack_kernel(
    Geom g, SortedPoints<T> sp, const T* __restrict__ horner, const T* __restrict__ c,
    T* __restrict__ fw, int64_t c_stride, int64_t fw_stride, T scale) {
  using T2 = typename Pair<T>::type;
  constexpr int LS = 24, L0 = 16 + W - 1, L1 = 16 + W - 1, L2 = TZ + W - 1;
  constexpr int PS = LS * L1;
  constexpr int plane = PS * L2;
  constexpr int PAD = kWave3Pad<W>;
  constexpr int SW = W <= 6 ? 6 : 8;                                  // staging row length
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* plane_re = reinterpret_cast<double*>(smem_raw);
  double* plane_im = plane_re + plane;
  T* stage_all = reinterpret_cast<T*>(plane_re + 2 * plane + PAD);
  const int s = blockIdx.x;
  if (s >= sp.seg_count[0]) return;
  const StackDesc d = stack_load(sp.segs, s);
  const StackColumn col = stack_column(g, d.col);
  const int slot = col.item * (int)gridDim.y + (int)blockIdx.y;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * plane + PAD; i += NW * 64) plane_re[i] = 0.0;
  const T2* cc = reinterpret_cast<const T2*>(c) + (int64_t)slot * c_stride;
  const int nc = g.ncoef;
  T* kxs = stage_all + wave * (CH * 2 * SW);   // [CH][SW]
  T* kys = kxs + CH * SW;                      // [CH][SW]
  const int dx = lane & 7, dy = lane >> 3;
  const bool active = dx < W && dy < W;
  const int cell = dy * LS + dx;
  const int o0 = col.t0 * 16, o1 = col.t1 * 16;
  T* out = fw + 2 * (int64_t)slot * fw_stride;
  __syncthreads();
  for (int i = 0; i < d.nz; ++i) {
    int p0 = d.p0, p1 = d.p1;
    if (p0 < 0) {
      const int t = stack_tile_index(g, col, d.z0 + i);
      p0 = sp.tile_start[t];
      p1 = sp.tile_start[t + 1];
    }
    const int npt = p1 - p0;
    const int share = (npt + NW - 1) / NW;
    const int wbeg = p0 + wave * share;
    const int wend = (wbeg + share < p1) ? wbeg + share : p1;
    // ---- the tile's points into the planes (the loop of spread_wave3_body, fp64 planes, both components)
    for (int base = wbeg; base < wend; base += CH) {
      const int j = base + lane;
      int off = 0;
      T kz[W];
      T cre = (T)0, cim = (T)0;
#pragma unroll
      for (int q = 0; q < W; ++q) kz[q] = (T)0;
      if (lane < CH) {
        T kx[W], ky[W];
#pragma unroll
        for (int q = 0; q < W; ++q) { kx[q] = (T)0; ky[q] = (T)0; }
        if (j < wend) {
          const PointView<T> rec = unpack_rec<T, 3>(sp.rec[j]);
          const T2 cv = cc[rec.idx];
          cre = cv.x * scale;
          cim = cv.y * scale;
          off = (int)(rec.loc & 1023) + (int)((rec.loc >> 10) & 1023) * LS + (int)((rec.loc >> 20) & 1023) * PS;
          T h0[8], h1[8], h2[8];
          horner8<T, 3>(horner, nc, rec.z0, rec.z1, rec.z2, h0, h1, h2);
#pragma unroll
          for (int q = 0; q < W; ++q) { kx[q] = h0[q]; ky[q] = h1[q]; kz[q] = h2[q]; }
        }
#pragma unroll
        for (int q = 0; q < W; ++q) {
          kxs[lane * SW + q] = kx[q];
          kys[lane * SW + q] = ky[q];
        }
      }
      int npts = wend - base;
      if (npts > CH) npts = CH;
      T kx_n = kxs[dx], ky_n = kys[dy];
      for (int q = 0; q < npts; ++q) {
        const T a = active ? kx_n * ky_n : (T)0;   // (lanes outside the patch add 0 at their natural patch address)
        const int qn = (q + 1 < npts) ? q + 1 : q;
        kx_n = kxs[qn * SW + dx];
        ky_n = kys[qn * SW + dy];
        const int o = __builtin_amdgcn_readlane(off, q) + cell;
        const T ar = a * bcast_lane(cre, q);
        const T ai = a * bcast_lane(cim, q);
        double* pr = plane_re + o;
        double* pi = plane_im + o;
#pragma unroll
        for (int dz = 0; dz < W; ++dz) {
          const T kzq = bcast_lane(kz[dz], q);
          lds_add(pr + dz * PS, (double)(ar * kzq));
          lds_add(pi + dz * PS, (double)(ai * kzq));
        }
      }
    }
    __syncthreads();
    // ---- tile d.z0 + i is complete in its first TZ planes (the last tile of the stack: in all of them)
    const bool last = i == d.nz - 1;
    const int o2 = (d.z0 + i) * TZ;
    const int nrows = (last ? L2 : TZ) * L1;
    const int e = lane < 2 * L0 ? lane : 2 * L0 - 1, a0 = e >> 1, comp = e & 1;
    const bool lane_on = lane < 2 * L0;
    const int gx = wrap1(o0 + a0, g.nf[0]);
    double* pl = comp ? plane_im : plane_re;
    for (int rho = wave; rho < nrows; rho += NW) {
      const int a2 = rho / L1, a1 = rho - a2 * L1;
      const int lrow = a2 * PS + a1 * LS + a0;
      const int64_t gbase = (int64_t)g.nf[0] * (wrap1(o1 + a1, g.nf[1]) + (int64_t)g.nf[1] * wrap1(o2 + a2, g.nf[2]));
      constexpr int CHAIN = (L2 + TZ - 1) / TZ;   // planes a2, a2 + TZ, ... of this (y, x) column
      double v[CHAIN];
#pragma unroll
      for (int m = 0; m < CHAIN; ++m) v[m] = (m == 0 || (!last && a2 + m * TZ < L2)) ? pl[lrow + m * TZ * PS] : 0.0;
      if (lane_on && v[0] != 0.0) glb_add(&out[2 * (gbase + gx) + comp], (T)v[0]);
      if (!last && lane_on) {
#pragma unroll
        for (int m = 0; m < CHAIN; ++m)
          if (a2 + m * TZ < L2) pl[lrow + m * TZ * PS] = m + 1 < CHAIN ? v[m + 1] : 0.0;
      }
    }
    if (!last) __syncthreads();
  }
}

// Fixed-point plans of the r01-r03 kernels (w <= 6 dense, w = 7 on depth-4 tiles): the subproblems of every tile with more
// than fx_max_subs of them, as a list for the persistent fp64-plane launches (run in set_points; list[0] = count, zeroed
// by the launcher)
__global__ __launch_bounds__(256) void crowded_list_kernel(Geom g, const int32_t* __restrict__ sub_start, int* __restrict__ list) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= g.ntiles) return;
  const int s0 = sub_start[t], k = sub_start[t + 1] - s0;
  if (k <= g.fx_max_subs) return;
  const int base = atomicAdd(&list[0], k);
  for (int j = 0; j < k; ++j) list[1 + base + j] = s0 + j;
}

// ---------------- interp: LDS tile, one thread per point, compile-time width

// A first version gathered 8 points per wavefront pass (lane = (point, dx),
// kernel values staged in LDS, 8-lane DPP reductions); PMC showed it
// instruction-issue bound (14 issue slots per point: staging, meta reads, DPP,
// scalar loop control; 2-D 309 us, 3-D 17.2 ms at M = 1e8). Replaces the
// reference's InterpSubproblem* kernels (nufft_plan.cu.cc:1041-1187, 1608-1804).
// Here a thread keeps its point's kernel values in registers and walks the
// W^RANK stencil with immediate LDS offsets: one ds_read_b64 + one packed FMA
// per cell, ~4.5 issue slots per point (2-D, W = 8). Bank conflicts between
// the 64 unrelated points of a wavefront remain (random cells), but the LDS
// pipe then does ~7 cycles per point against ~14 issue-cycles before.
template <typename T, int NDIM, int W>
__device__ __forceinline__ void hornerW(const T* __restrict__ tab, int nc, T z0, T z1, T z2,
                                        T (&k0)[W], T (&k1)[W], T (&k2)[W]) {
  if (nc <= kFixedCoef) {
#pragma unroll
    for (int q = 0; q < W; ++q) {
      const T t = tab[(kFixedCoef - 1) * kMaxW + q];
      k0[q] = t; k1[q] = t; k2[q] = t;
    }
#pragma unroll
    for (int k = kFixedCoef - 2; k >= 0; --k) {
#pragma unroll
      for (int q = 0; q < W; ++q) {
        const T t = tab[k * kMaxW + q];
        k0[q] = fma(k0[q], z0, t);
        if (NDIM > 1) k1[q] = fma(k1[q], z1, t);
        if (NDIM > 2) k2[q] = fma(k2[q], z2, t);
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < W; ++q) {
      const T t = tab[(nc - 1) * kMaxW + q];
      k0[q] = t; k1[q] = t; k2[q] = t;
    }
    for (int k = nc - 2; k >= 0; --k) {
#pragma unroll
      for (int q = 0; q < W; ++q) {
        const T t = tab[k * kMaxW + q];
        k0[q] = fma(k0[q], z0, t);
        if (NDIM > 1) k1[q] = fma(k1[q], z1, t);
        if (NDIM > 2) k2[q] = fma(k2[q], z2, t);
      }
    }
  }
}

// Threads per workgroup: 256 in 2-D (12.5 KB tiles, the thread limit of the CU binds), 512
// in 3-D, where a 46 KB tile lets only three workgroups share a CU.
// The 64 x 64 tiles of 2-D float type-2 plans (41 KB: three workgroups per CU) take 512 as well: 12 -> 24 waves per CU,
// config 3 interp 256 -> 241 us (r03; 1024 threads: 260).
#ifndef NUFFT_INTERP_EXP   // (experiment builds, tools/interp_loop_experiment.sh: 1 no LDS reads, 2 no result stores, 4 no tile load,
#define NUFFT_INTERP_EXP 0 //  8 results stored in sorted order -- wrong results, timing only)
#endif
template <int RANK> constexpr int kInterpThreads = RANK > 2 ? 512 : 256;
constexpr double kSplitPointsPerTile = 64.0;   // (r06; see interp_point_kernel SPLIT and profiles/r06_interp_split_ab.txt)
// STACK (r06; 3-D, 16 x 16 x TZ tiles, double precision): the workgroup walks a STACK of tiles consecutive in z
// (stack_plan_kernel, nufft_dense3.hip) instead of one subproblem -- the w - 1 planes a tile shares with the next one
// move down in LDS and only TZ new planes are read. On 16-byte cells the tiles are 16 x 16 x 4 and tile + halo is 5.7 x
// the tile, read per subproblem by the one workgroup a CU holds (93 KB): 7.8 ms for 256^3 modes at M = 1e7 against 1.55 ms
// in float (profiles/r06_c128_before.txt). (The float kernel's stack form lost to the flat loader at three workgroups
// per CU: EXPERIMENTS.md 11.7.)
// SPLIT (r06; 3-D): EIGHT lanes per point, lane s summing z plane s of the stencil (w x w cells), the partial sums
// combined over the eight lanes. A thread per point leaves a tile with 20-150 points on one or two waves of the
// workgroup, each walking w^3 dependent cell reads and FMAs alone (512 at w = 8: ~3.5 us in double) -- below ~0.3 points
// per cell that latency, not the tile read, was the kernel (complex128, 256^3 modes: interp 7.2-7.8 ms from M = 3e6 to
// 1e7, unchanged by halving the reads with stacks: profiles/r06_c128_interp_stack_ab.txt).
template <typename T, int RANK, int W, int NTHREADS = kInterpThreads<RANK>, bool STACK = false, bool SPLIT = false>
__global__ __launch_bounds__(NTHREADS) void interp_point_kernel(
    Geom g, SortedPoints<T> sp, const T* __restrict__ horner, T* __restrict__ c,
    const T* __restrict__ fw, int64_t c_stride, int64_t fw_stride, T scale) {
  static_assert(!STACK || RANK == 3, "stacks: 3-D");
  static_assert(!SPLIT || RANK == 3, "eight lanes per point: 3-D");
  using T2 = typename Pair<T>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int LS = g.lstride;
  const int L0 = g.ldim[0], L1 = g.ldim[1];
  const int L2 = RANK > 2 ? g.ldim[2] : 1;
  const int PS = LS * L1;
  T2* tile = reinterpret_cast<T2*>(smem_raw);
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
  const int o0 = t0 * g.tile[0], o1 = t1 * g.tile[1];
  int o2 = t2 * g.tile[2];
  const T2* in = reinterpret_cast<const T2*>(fw) + (int64_t)slot * fw_stride;
  constexpr int NT = NTHREADS;
  // Tile rows (L0 <= 39 cells: one lane per cell) are fetched kRowBatch at a time: the
  // loads of a batch are issued back to back on clamped addresses and only then stored
  // to LDS. One load - wait - store per row exposed an L2 latency per row (10 rows per
  // wave in 2-D, 35 in 3-D) at the head of every workgroup.
  constexpr int kRowBatch = 8;
  const int nrows = L1 * L2;
  // 3-D tiles of 16 x 16 cells in x and y (r05): the cells are dealt to the threads one after the other instead of a
  // row per wave -- a row is 16 + W - 1 <= 23 cells, a third of a wave, and a wave walked its ~43 rows in 5-6 batches
  // of loads one after the other; this way a thread's ~15 cells are two batches (the row length is compile-time here:
  // the divisions are multiplications). 3-D type 2, 256^3 modes, w = 8, interp stage: M = 3e6 2.40 -> 1.41 ms, 1e7
  // 2.58 -> 1.58, 3e7 3.99 -> 3.10, 1e8 7.32 -> 6.77; w = 6, 1e7: 1.78 -> 1.03. (The 2-D tiles -- 39- and 71-cell rows, the
  // latter in two sweeps -- measured the same either way, config 3 221 against 221-227 us: left on rows.)
  auto load_flat = [&](auto lc, int plane0) {   // planes [plane0, L2) of the tile at (o0, o1, o2)
    constexpr int LC = decltype(lc)::value;
    const int ncell = LC * LC * (L2 - plane0);
    for (int e0 = tid; e0 < ncell; e0 += kRowBatch * NT) {
      T2 v[kRowBatch];
      int lofs[kRowBatch];
#pragma unroll
      for (int u = 0; u < kRowBatch; ++u) {
        const int e = e0 + u * NT;
        const int ec = e < ncell ? e : ncell - 1;
        const int a2r = RANK > 2 ? ec / (LC * LC) : 0, r = ec - a2r * (LC * LC);
        const int a2 = a2r + plane0;
        const int a1 = r / LC, a0 = r - a1 * LC;
        const int g2 = RANK > 2 ? wrap1(o2 + a2, g.nf[2]) : 0;
        v[u] = in[(int64_t)g.nf[0] * (wrap1(o1 + a1, g.nf[1]) + (int64_t)g.nf[1] * g2) + wrap1(o0 + a0, g.nf[0])];
        lofs[u] = e < ncell ? a2 * PS + a1 * LS + a0 : -1;
      }
#pragma unroll
      for (int u = 0; u < kRowBatch; ++u)
        if (lofs[u] >= 0) tile[lofs[u]] = v[u];
    }
  };
  T2* cc = reinterpret_cast<T2*>(c) + (int64_t)slot * c_stride;
  const int nc = g.ncoef;
  // index of the first record a thread reads of a tile's points (clamped)
  auto first_index = [&](int q0, int q1) {
    const int j = SPLIT ? q0 + (tid >> 3) : q0 + tid;
    return j < q1 ? j : (q1 > q0 ? q1 - 1 : 0);   // (an empty tile: any valid record)
  };
  // the points [p0, p1) of the tile in LDS; `first` = sp.rec[first_index(p0, p1)], loaded by the caller (a tile ahead on stacks)
  auto do_points = [&](int p0, int p1, const Rec<T> first) {
    if (p1 <= p0) return;
    if constexpr (SPLIT) {
      const int sl = tid & 7;                  // the z plane of the stencil this lane sums
      const int sq = sl < W ? sl : 0;
      constexpr int NP = NT / 8;               // points per pass of the workgroup
      for (int j0 = p0; j0 < p1; j0 += NP) {
        const int j = j0 + (tid >> 3);
        const bool live = j < p1;
        const PointView<T> rec = unpack_rec<T, RANK>(j0 == p0 ? first : sp.rec[live ? j : p1 - 1]);
        // tap sl of the three kernel polynomials; the x and y taps of the other seven lanes come by shuffle
        T k0 = horner[(nc - 1) * kMaxW + sq], k1 = k0, k2 = k0;
        for (int k = nc - 2; k >= 0; --k) {
          const T t = horner[k * kMaxW + sq];
          k0 = fma(k0, rec.z0, t);
          k1 = fma(k1, rec.z1, t);
          k2 = fma(k2, rec.z2, t);
        }
        if (sl >= W) k2 = (T)0;                // (lanes past the width re-read plane 0 with weight 0)
        T kx[W], ky[W];
#pragma unroll
        for (int q = 0; q < W; ++q) {
          kx[q] = __shfl(k0, q, 8);
          ky[q] = __shfl(k1, q, 8);
        }
        const T2* tp = tile + (int)(rec.loc & 1023) + (int)((rec.loc >> 10) & 1023) * LS +
                       ((int)((rec.loc >> 20) & 1023) + sq) * PS;
        T pre = (T)0, pim = (T)0;
#pragma unroll
        for (int dy = 0; dy < W; ++dy) {
          const T2* row = tp + dy * LS;
          T rre = (T)0, rim = (T)0;
#pragma unroll
          for (int dx = 0; dx < W; ++dx) {
            const T2 v = lds_cell(row + dx);
            rre = fma(kx[dx], v.x, rre);
            rim = fma(kx[dx], v.y, rim);
          }
          pre = fma(ky[dy], rre, pre);
          pim = fma(ky[dy], rim, pim);
        }
        T sre = k2 * pre, sim = k2 * pim;
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
          sre += __shfl_xor(sre, o, 8);
          sim += __shfl_xor(sim, o, 8);
        }
        if (sl == 0 && live) {
          T2 out;
          out.x = sre * scale;
          out.y = sim * scale;
          cc[rec.idx] = out;
        }
      }
      return;
    }

  // the next record is requested (on a clamped index, outside any branch) before this
    // point's ~300 dependent instructions, so its HBM latency is off the critical path
    Rec<T> raw = first;
    for (int j = p0 + tid; j < p1; j += NT) {
      const PointView<T> rec = unpack_rec<T, RANK>(raw);
      raw = sp.rec[j + NT < p1 ? j + NT : p1 - 1];
      T kx[W], ky[W], kz[W];
      hornerW<T, RANK, W>(horner, nc, rec.z0, rec.z1, rec.z2, kx, ky, kz);
      const T2* tp = tile + (int)(rec.loc & 1023) + (int)((rec.loc >> 10) & 1023) * LS +
                     (RANK > 2 ? (int)((rec.loc >> 20) & 1023) * PS : 0);
      T sre = (T)0, sim = (T)0;
      // (3-D: three z-planes per unrolled body. Fully unrolled -- 216 to 512 cell reads -- the kernel held 114 VGPRs and
      // up: two workgroups per CU at w = 6, spills at w = 7, 8; this way 75-79: M = 3e7 at w = 8 7.7 -> 4.0 ms, r03)
#pragma unroll 3
      for (int dz = 0; dz < (RANK > 2 ? W : 1); ++dz) {
        T pre = (T)0, pim = (T)0;
#pragma unroll
        for (int dy = 0; dy < W; ++dy) {
          const T2* row = tp + dz * PS + dy * LS;
          T rre = (T)0, rim = (T)0;
#pragma unroll
          for (int dx = 0; dx < W; ++dx) {
#if NUFFT_INTERP_EXP & 1
            T2 v; v.x = (T)(dx + dy); v.y = (T)dz;
#else
            const T2 v = lds_cell(row + dx);
#endif
            rre = fma(kx[dx], v.x, rre);
            rim = fma(kx[dx], v.y, rim);
          }
          pre = fma(ky[dy], rre, pre);
          pim = fma(ky[dy], rim, pim);
        }
        if (RANK > 2) {
          sre = fma(kz[dz], pre, sre);
          sim = fma(kz[dz], pim, sim);
        } else {
          sre = pre;
          sim = pim;
        }
      }
      T2 out;
      out.x = sre * scale;
      out.y = sim * scale;
#if NUFFT_INTERP_EXP & 2
      if (out.x == (T)123.456) cc[rec.idx] = out;
#elif NUFFT_INTERP_EXP & 8
      cc[j] = out;   // (in sorted order: coalesced)
#else
      cc[rec.idx] = out;
#endif
    }
  };
  if constexpr (STACK) {
    // ---- a stack of tiles, pipelined: every tile's point range is read up front, the next tile's new planes and
    // first records are requested BEFORE this tile's points are interpolated (r05's lesson from the float stack kernels:
    // a sparse tile's time is the latency of its dependent loads -- range, then record, then planes -- not their bytes)
    constexpr int LC = 16 + W - 1;
    constexpr int TZ = (sizeof(T) == 4 || W <= 6) ? 8 : 4;  // (the host launches this form on such tiles only)
    constexpr int NPF = (LC * LC * TZ + NT - 1) / NT;       // cells of a tile's new planes per thread
    constexpr int kRng = 64;
    __shared__ int rng[2 * kRng];
    if (sd.p0 < 0) {
      for (int i = tid; i < sd.nz && i < kRng; i += NT) {
        const int t = stack_tile_index(g, scol, sd.z0 + i);
        rng[2 * i] = sp.tile_start[t];
        rng[2 * i + 1] = sp.tile_start[t + 1];
      }
    }
    auto range_of = [&](int i, int* q0, int* q1) {
      if (sd.p0 >= 0) { *q0 = sd.p0; *q1 = sd.p1; return; }   // a piece: one tile, its own points
      if (i < kRng) { *q0 = rng[2 * i]; *q1 = rng[2 * i + 1]; return; }
      const int t = stack_tile_index(g, scol, sd.z0 + i);
      *q0 = sp.tile_start[t];
      *q1 = sp.tile_start[t + 1];
    };
    o2 = sd.z0 * TZ;
    load_flat(std::integral_constant<int, LC>(), 0);
    __syncthreads();
    int q0, q1;
    range_of(0, &q0, &q1);
    Rec<T> first = sp.rec[first_index(q0, q1)];
    for (int ti = 0; ti < sd.nz; ++ti) {
      const bool more = ti + 1 < sd.nz;
      const int c0 = q0, c1 = q1;
      const Rec<T> first_now = first;
      // the next tile's new planes, NPF cells per thread, in NAMED registers (as an array indexed by an unrolled loop they
      // went to scratch memory: 80-128 bytes per lane); requested unconditionally -- behind the last tile the planes it
      // already holds are read again and dropped
      T2 pf0, pf1, pf2, pf3, pf4, pf5, pf6, pf7, pf8, pf9;
      static_assert(NPF <= 10, "prefetch registers");
      const int tn = more ? ti + 1 : ti;
      range_of(tn, &q0, &q1);
      first = sp.rec[first_index(q0, q1)];
      const int o2n = (sd.z0 + tn) * TZ + (L2 - TZ);   // first new plane of the next tile, fine-grid z before wrapping
      auto pf_src = [&](int u) {
        const int e = tid + u * NT;
        const int ec = e < LC * LC * TZ ? e : LC * LC * TZ - 1;
        const int a2 = ec / (LC * LC), r = ec - a2 * (LC * LC);
        const int a1 = r / LC, a0 = r - a1 * LC;
        return in[(int64_t)g.nf[0] * (wrap1(o1 + a1, g.nf[1]) + (int64_t)g.nf[1] * wrap1(o2n + a2, g.nf[2])) + wrap1(o0 + a0, g.nf[0])];
      };
#define NUFFT_PF_LOAD(u) if constexpr (u < NPF) pf##u = pf_src(u); else pf##u = T2();
      NUFFT_PF_LOAD(0) NUFFT_PF_LOAD(1) NUFFT_PF_LOAD(2) NUFFT_PF_LOAD(3) NUFFT_PF_LOAD(4) NUFFT_PF_LOAD(5) NUFFT_PF_LOAD(6) NUFFT_PF_LOAD(7)
      NUFFT_PF_LOAD(8) NUFFT_PF_LOAD(9)
#undef NUFFT_PF_LOAD
      do_points(c0, c1, first_now);
      if (!more) break;
      __syncthreads();   // every thread is done with this tile's planes
      // planes TZ .. L2 - 1 move down by TZ: one thread per (y, x) column and residue of the plane index, upwards
      for (int e = tid; e < LC * LC * TZ; e += NT) {
        const int a2 = e / (LC * LC), r = e - a2 * (LC * LC);
        const int a1 = r / LC, a0 = r - a1 * LC;
        T2* col = tile + a1 * LS + a0;
        for (int q = a2; q + TZ < L2; q += TZ) col[q * PS] = col[(q + TZ) * PS];
      }
      __syncthreads();   // (the new planes land where the moved ones were read)
      auto pf_dst = [&](int u, const T2& v) {
        const int e = tid + u * NT;
        if (e < LC * LC * TZ) {
          const int a2 = e / (LC * LC), r = e - a2 * (LC * LC);
          const int a1 = r / LC, a0 = r - a1 * LC;
          tile[(L2 - TZ + a2) * PS + a1 * LS + a0] = v;
        }
      };
#define NUFFT_PF_STORE(u) if constexpr (u < NPF) pf_dst(u, pf##u);
      NUFFT_PF_STORE(0) NUFFT_PF_STORE(1) NUFFT_PF_STORE(2) NUFFT_PF_STORE(3) NUFFT_PF_STORE(4) NUFFT_PF_STORE(5) NUFFT_PF_STORE(6) NUFFT_PF_STORE(7)
      NUFFT_PF_STORE(8) NUFFT_PF_STORE(9)
#undef NUFFT_PF_STORE
      __syncthreads();
    }
    return;
  }
  if (NUFFT_INTERP_EXP & 4) { }
  else if (RANK > 2 && g.tile[0] == 16 && g.tile[1] == 16) load_flat(std::integral_constant<int, 16 + W - 1>(), 0);
  else
  // (rows of more than 64 cells -- the 64 x 64 tiles of 2-D type-2 plans, 71 cells with the
  // halo -- take a second sweep for the remaining columns)
  for (int c0 = 0; c0 < L0; c0 += 64) {
    const int col = c0 + lane;
    const int a0c = col < L0 ? col : L0 - 1;
    const int64_t gx = wrap1(o0 + a0c, g.nf[0]);
    for (int rb = wave; rb < nrows; rb += kRowBatch * (NT / 64)) {
      T2 v[kRowBatch];
      int lofs[kRowBatch];
#pragma unroll
      for (int u = 0; u < kRowBatch; ++u) {
        const int row = rb + u * (NT / 64);
        const int rc = row < nrows ? row : nrows - 1;
        const int a2 = RANK > 2 ? rc / L1 : 0;
        const int a1 = rc - a2 * L1;
        const int g1 = wrap1(o1 + a1, g.nf[1]);
        const int g2 = RANK > 2 ? wrap1(o2 + a2, g.nf[2]) : 0;
        v[u] = in[(int64_t)g.nf[0] * (g1 + (int64_t)g.nf[1] * g2) + gx];
        lofs[u] = row < nrows ? a2 * PS + a1 * LS + col : -1;
      }
#pragma unroll
      for (int u = 0; u < kRowBatch; ++u)
        if (lofs[u] >= 0 && col < L0) tile[lofs[u]] = v[u];
    }
  }
  __syncthreads();
  do_points(p0, p1, sp.rec[first_index(p0, p1)]);
}

// ------------------------------------------------ interp: straight from the caller's points (r06)
//
// Small type-2 calls through the one-call entry: no sort at all. A thread folds its point (the sort kernels' own fold),
// evaluates the kernel and gathers its w^rank fine cells from global memory (L2 / Infinity Cache). A call of the size of
// the reference harness's first case (2-D, 256^2 modes, M = 2e5) spends 25 of its 50 us sorting points it interpolates once
// (four launches of 4-8 us: EXPERIMENTS.md 11.4); below ~1e5 points the gather is cheaper than that (12.9). Results leave
// in the caller's order: coalesced stores.
template <typename T, int RANK>
__global__ __launch_bounds__(256) void interp_direct_kernel(Geom g, PointsIn in, const T* __restrict__ horner, T* __restrict__ c,
                                                            const T* __restrict__ fw, int64_t c_stride, int64_t fw_stride,
                                                            T scale) {
  using T2 = typename Pair<T>::type;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= in.M) return;
  Rec<T> r;
  bool bad = false;
  const int tile = fold_point<T>(g, in, i, &r, &bad);
  int t0, t1, t2;
  tile_coords(g, tile, &t0, &t1, &t2);
  const PointView<T> pv = {r.loc, r.z0, r.z1, r.z2, 0};
  T k0[8], k1[8], k2[8];
  horner8<T, RANK>(horner, g.ncoef, pv.z0, pv.z1, pv.z2, k0, k1, k2);
  const int b0 = t0 * g.tile[0] + (int)(pv.loc & 1023);
  const int b1 = t1 * g.tile[1] + (int)((pv.loc >> 10) & 1023);
  const int b2 = RANK > 2 ? t2 * g.tile[2] + (int)((pv.loc >> 20) & 1023) : 0;
  int gx[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) gx[q] = wrap1(b0 + q, g.nf[0]);
  const int w = g.w;
  {
    const int y = (int)blockIdx.y;   // transform of the batch
    const T2* inp = reinterpret_cast<const T2*>(fw) + (int64_t)y * fw_stride;
    T sre = (T)0, sim = (T)0;
    for (int dz = 0; dz < (RANK > 2 ? w : 1); ++dz) {
      const int64_t zoff = RANK > 2 ? (int64_t)wrap1(b2 + dz, g.nf[2]) * g.nf[1] : 0;
      T pre = (T)0, pim = (T)0;
      for (int dy = 0; dy < w; ++dy) {
        const T2* row = inp + (zoff + wrap1(b1 + dy, g.nf[1])) * (int64_t)g.nf[0];
        T rre = (T)0, rim = (T)0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const T2 v = row[gx[q]];          // (taps beyond the width are exactly 0: the cell is read and ignored)
          rre = fma(k0[q], v.x, rre);
          rim = fma(k0[q], v.y, rim);
        }
        pre = fma(k1[dy], rre, pre);
        pim = fma(k1[dy], rim, pim);
      }
      if (RANK > 2) { sre = fma(k2[dz], pre, sre); sim = fma(k2[dz], pim, sim); }
      else { sre = pre; sim = pim; }
    }
    T2 out;
    out.x = sre * scale;
    out.y = sim * scale;
    (reinterpret_cast<T2*>(c) + (int64_t)y * c_stride)[i] = out;
  }
}

// ------------------------------------------------ interp: generic tile path

// One workgroup per subproblem, one thread per point, gathering w^rank fine
// cells straight from global memory (tile-sorted points keep the working set
// in the XCD's L2).
template <typename T, int RANK>
__global__ __launch_bounds__(kBlock) void interp_tile_generic_kernel(
    Geom g, SortedPoints<T> sp, const T* __restrict__ horner, T* __restrict__ c,
    const T* __restrict__ fw, int64_t c_stride, int64_t fw_stride, T scale) {
  int tb, p0, p1, slot;
  if (!locate_subproblem(g, sp.tile_start, sp.sub_start, blockIdx.x, &tb, &p0, &p1, &slot)) return;
  const int w = g.w, nc = g.ncoef;
  int t0, t1, t2;
  tile_coords(g, tb, &t0, &t1, &t2);
  const int o0 = t0 * g.tile[0], o1 = t1 * g.tile[1], o2 = t2 * g.tile[2];
  const T* in = fw + 2 * (int64_t)slot * fw_stride;
  T* cc = c + 2 * (int64_t)slot * c_stride;
  for (int j = p0 + (int)threadIdx.x; j < p1; j += kBlock) {
    const PointView<T> rec = unpack_rec<T, RANK>(sp.rec[j]);
    const uint32_t loc = rec.loc;
    const int idx = rec.idx;
    T kx[kMaxW];
    int gx[kMaxW];
    const T z0 = rec.z0;
    const int b0 = o0 + (int)(loc & 1023);
#pragma unroll
    for (int q = 0; q < kMaxW; ++q) {
      kx[q] = (q < w) ? horner_cell(horner, nc, q, z0) : (T)0;
      gx[q] = (b0 + q) % g.nf[0];
    }
    T sre = 0, sim = 0;
    if (RANK == 1) {
#pragma unroll
      for (int q = 0; q < kMaxW; ++q)
        if (q < w) {
          sre = fma(in[2 * (int64_t)gx[q]], kx[q], sre);
          sim = fma(in[2 * (int64_t)gx[q] + 1], kx[q], sim);
        }
    } else if (RANK == 2) {
      const int b1 = o1 + (int)((loc >> 10) & 1023);
      const T z1 = rec.z1;
      for (int dy = 0; dy < w; ++dy) {
        const T ky = horner_cell(horner, nc, dy, z1);
        const int64_t ro = (int64_t)g.nf[0] * ((b1 + dy) % g.nf[1]);
        T lre = 0, lim = 0;
#pragma unroll
        for (int q = 0; q < kMaxW; ++q)
          if (q < w) {
            lre = fma(in[2 * (ro + gx[q])], kx[q], lre);
            lim = fma(in[2 * (ro + gx[q]) + 1], kx[q], lim);
          }
        sre = fma(ky, lre, sre);
        sim = fma(ky, lim, sim);
      }
    } else {
      const int b1 = o1 + (int)((loc >> 10) & 1023);
      const int b2 = o2 + (int)((loc >> 20) & 1023);
      const T z1 = rec.z1;
      const T z2 = rec.z2;
      for (int dz = 0; dz < w; ++dz) {
        const T kz = horner_cell(horner, nc, dz, z2);
        const int64_t zo = (int64_t)g.nf[1] * ((b2 + dz) % g.nf[2]);
        for (int dy = 0; dy < w; ++dy) {
          const T kyz = kz * horner_cell(horner, nc, dy, z1);
          const int64_t ro = (int64_t)g.nf[0] * (zo + (b1 + dy) % g.nf[1]);
          T lre = 0, lim = 0;
#pragma unroll
          for (int q = 0; q < kMaxW; ++q)
            if (q < w) {
              lre = fma(in[2 * (ro + gx[q])], kx[q], lre);
              lim = fma(in[2 * (ro + gx[q]) + 1], kx[q], lim);
            }
          sre = fma(kyz, lre, sre);
          sim = fma(kyz, lim, sim);
        }
      }
    }
    cc[2 * (int64_t)idx] = sre * scale;
    cc[2 * (int64_t)idx + 1] = sim * scale;
  }
}

// -------------------------------------------------------------- deconvolve

// dir 1 (type-1 step 3): f[k] = fw[k mod nf] * rf0[|k0|] rf1[|k1|] rf2[|k2|],
//   one thread per output mode, CMCL order (index 0 = most negative mode):
//   fuses fftshift + truncation nf -> N + division by the kernel's Fourier
//   series (reference Deconvolve{1,2,3}DKernel nufft_plan.cu.cc:326-379;
//   CPU deconvolve_*d nufft_plan.cc:729-881).
// dir 2 (type-2 step 1): one thread per FINE cell writes either the amplified
//   mode or zero -- the zero-fill of the whole fine batch the reference does
//   with a separate memset (nufft_plan.cu.cc:2855-2858) is fused in.
// rf* hold RECIPROCALS of the Fourier series (computed in double on the host).
template <typename T, bool FLAT>
__global__ __launch_bounds__(256) void deconvolve_kernel(Geom g, int dir, int nbatch, T* __restrict__ f,
                                                         T* __restrict__ fw, const T* __restrict__ rf0,
                                                         const T* __restrict__ rf1,
                                                         const T* __restrict__ rf2) {
  // grid: x over the fastest dimension, y = row (second dimension), z = slab * batch;
  // no per-element division
  const int N0 = g.nmodes[0], N1 = g.nmodes[1], N2 = g.nmodes[2];
  const int nf0 = g.nf[0], nf1 = g.nf[1], nf2 = g.nf[2];
  const int64_t ntot = (int64_t)N0 * N1 * N2, nftot = (int64_t)nf0 * nf1 * nf2;
  const int d1 = dir == 1 ? N1 : nf1;
  const int d2 = dir == 1 ? N2 : nf2;
  int b, i2, i1;
  if (FLAT) {   // grids beyond the 65535 limit of gridDim.y / .z: rows (i1, i2, b) flattened along y
    const int64_t row = (int64_t)blockIdx.y + (int64_t)gridDim.y * blockIdx.z;
    if (row >= (int64_t)d1 * d2 * nbatch) return;
    const int64_t q = row / d1;
    i1 = (int)(row - q * d1);
    b = (int)(q / d2);
    i2 = (int)(q - (int64_t)b * d2);
  } else {
    b = blockIdx.z / d2;
    i2 = blockIdx.z - b * d2;
    i1 = blockIdx.y;
  }
  const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
  T2_t<T>* fb = reinterpret_cast<T2_t<T>*>(f) + (int64_t)b * ntot;
  T2_t<T>* fwb = reinterpret_cast<T2_t<T>*>(fw) + (int64_t)b * nftot;
  if (dir == 1) {
    if (i0 >= N0) return;
    const int k0 = i0 - N0 / 2, k1 = i1 - N1 / 2, k2 = i2 - N2 / 2;
    const int w0 = k0 >= 0 ? k0 : nf0 + k0;
    const int w1 = k1 >= 0 ? k1 : nf1 + k1;
    const int w2 = k2 >= 0 ? k2 : nf2 + k2;
    T r = rf0[k0 < 0 ? -k0 : k0];
    if (g.rank > 1) r *= rf1[k1 < 0 ? -k1 : k1];
    if (g.rank > 2) r *= rf2[k2 < 0 ? -k2 : k2];
    const T2_t<T> v = fwb[w0 + (int64_t)nf0 * (w1 + (int64_t)nf1 * w2)];
    T2_t<T> o;
    o.x = v.x * r;
    o.y = v.y * r;
    fb[i0 + (int64_t)N0 * (i1 + (int64_t)N1 * i2)] = o;
  } else {
    if (i0 >= nf0) return;
    // kept modes: k in [-(N/2), (N-1)/2]
    const int k0 = i0 <= (N0 - 1) / 2 ? i0 : i0 - nf0;
    const int k1 = i1 <= (N1 - 1) / 2 ? i1 : i1 - nf1;
    const int k2 = i2 <= (N2 - 1) / 2 ? i2 : i2 - nf2;
    const bool keep = k0 >= -(N0 / 2) && k1 >= -(N1 / 2) && k2 >= -(N2 / 2);
    T2_t<T> o;
    o.x = (T)0;
    o.y = (T)0;
    if (keep) {
      T r = rf0[k0 < 0 ? -k0 : k0];
      if (g.rank > 1) r *= rf1[k1 < 0 ? -k1 : k1];
      if (g.rank > 2) r *= rf2[k2 < 0 ? -k2 : k2];
      const T2_t<T> v = fb[(k0 + N0 / 2) + (int64_t)N0 * ((k1 + N1 / 2) + (int64_t)N1 * (k2 + N2 / 2))];
      o.x = v.x * r;
      o.y = v.y * r;
    }
    fwb[i0 + (int64_t)nf0 * (i1 + (int64_t)nf1 * i2)] = o;
  }
}

// ------------------------------------------------------------------ permute

struct PermuteArgs {
  int ndim;
  int64_t shape[12];
  int64_t stride[12];
  int64_t total;
};

template <typename V>
__global__ __launch_bounds__(256) void permute_kernel(const V* __restrict__ src, V* __restrict__ dst,
                                                      PermuteArgs a) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.total) return;
  int64_t rem = i, so = 0;
  for (int d = a.ndim - 1; d >= 0; --d) {
    const int64_t q = rem % a.shape[d];
    rem /= a.shape[d];
    so += q * a.stride[d];
  }
  dst[i] = src[so];
}

}  // namespace

// ------------------------------------------------------------------ launchers

static inline unsigned blocks_for(int64_t n, int per) { return (unsigned)((n + per - 1) / per); }

template <typename K>
static hipError_t ensure_lds(K kernel, size_t bytes) {
  if (bytes > 64 * 1024)
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  return hipSuccess;
}

// The HIP runtime loads this library's device code lazily, at the first kernel launch
// (deferred loading). With that default the FIRST launch of a fresh process faulted in
// 5-15 % of runs on ROCm 7.2 / gfx950 ("Memory access fault ... write access to a
// read-only page", at addresses unrelated to any buffer of ours), as soon as the code
// object grew past ~1.25 MB; never with HIP_ENABLE_DEFERRED_LOADING=0, never on a later
// launch (r01: 5/40 against 0/40 runs; serialised launches put the fault inside the very
// first kernel). So the plan forces the load here -- a function-attribute query makes the
// runtime build the module for the device -- and waits for the device before anything of
// ours is launched.
#ifdef NUFFT_HIP_PHASE_LOG
extern "C" int nufft_hip_debug_phase_log(unsigned long long* dst, int n) {   // experiment build only
  if (n > kPhaseLogWgs * kPhaseSlots) n = kPhaseLogWgs * kPhaseSlots;
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_phase_log), sizeof(unsigned long long) * (size_t)n);
}
#endif
hipError_t preload_device_code() {
#ifdef NUFFT_HIP_NO_PRELOAD   // build macro of tools/first_launch_experiment.sh: does the fault still reproduce?
  return hipSuccess;
#endif
  static std::mutex mu;
  static bool done[64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lk(mu);
  if (dev >= 0 && dev < 64 && done[dev]) return hipSuccess;
  hipFuncAttributes attr;
  e = hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(scan_tiles_kernel));
  if (e != hipSuccess) return e;
  e = hipDeviceSynchronize();
  if (e != hipSuccess) return e;
  if (dev >= 0 && dev < 64) done[dev] = true;
  return hipSuccess;
}

// Below this many points (all sets together) the sort runs many small workgroups and the plain
// scatter; from here on blocks of >= 4096 points and, tile count permitting, the staged scatter
// (measured r02 on 512^2 grids: M = 2e6 0.141 -> 0.134 ms per transform in the small form, M = 4e6
// 0.190 -> 0.201).
constexpr int64_t kSmallSortPoints = (int64_t)1 << 21;
int sort_blocks(const Geom& g, int64_t M, int64_t* per_block) {
  // at most ~512 workgroups (longer per-tile runs per workgroup => better write
  // combining in the scatter; measured r01), at least 4096 points each; a workgroup
  // stays inside one point set, so the count is per set
  const int maxblk = 512;   // (tools/sweep_sort.py, r01: 256 and 1024 measured slower)
  const int items = g.nitems > 1 ? g.nitems : 1;
  const int64_t m_item = M / items;
  // (small point sets: 1024 per workgroup, so that a few hundred thousand points still spread over the
  // chip -- the reference benchmark's M = 2e5 cases 62 -> 52 us per transform with the plain scatter)
  const int minpb = M < kSmallSortPoints ? 1024 : 4096;
  int64_t pb = (M + maxblk - 1) / maxblk;
  if (pb < minpb) pb = minpb;
  pb += pb & 1;   // even: the paired 16-byte loads of interleaved 2-D float points start on a pair
  int64_t bpi = (m_item + pb - 1) / pb;
  if (bpi < 1) bpi = 1;
  // colscan_kernel scans at most kScanGroups * kScanRowsMax rows
  while (bpi * items > kScanGroups * kScanRowsMax && bpi > 1) { pb *= 2; bpi = (m_item + pb - 1) / pb; }
  *per_block = pb;
  return (int)bpi;
}

// The two-level sort exists for this plan (tiles numbered by super-tiles) and the point count makes it pay
// (six launches instead of four; options.tuning SORT2_ON: always)
constexpr int64_t kSort2MinPoints = (int64_t)3 << 19;   // (tools/sort2_ab.py: level at 1e6 points, 20 % ahead at 2e6)
static bool sort2_wanted(const Geom& g, int64_t M) {
  if (g.sup_shift[0] + g.sup_shift[1] + g.sup_shift[2] == 0 || g.nitems > 1 || M >= ((int64_t)1 << 31)) return false;
  return (g.tuning & NUFFT_HIP_TUNE_SORT2_ON) || M >= kSort2MinPoints;
}
// Geometry of the first level: the super-tiles as tiles
static Geom coarse_geom(const Geom& g) {
  Geom c = g;
  int nt = 1;
  for (int d = 0; d < 3; ++d) {
    c.tile[d] = g.tile[d] << g.sup_shift[d];
    c.tile_shift[d] = g.tile_shift[d] + g.sup_shift[d];
    c.ntile[d] = c.nsup[d] = g.nsup[d];
    c.sup_shift[d] = 0;
    nt *= c.ntile[d];
  }
  c.ntiles = c.ntiles_item = nt;
  return c;
}

// 0: LDS histogram, 32-bit counters; 1: LDS histogram, packed 16-bit counters
// (workgroups capped at 65535 points); 2: global counters; 3: two levels (super-tiles, then tiles).
int sort_mode(const Geom& g, int64_t M) {
  const int items = g.nitems > 1 ? g.nitems : 1;
  if (sort2_wanted(g, M)) return 3;
  if (g.ntiles <= kMaxLdsTiles && items <= kScanGroups * kScanRowsMax) return 0;
  if ((int64_t)g.ntiles <= (int64_t)kMaxRanges16 * kMaxLds16Tiles) {
    int64_t pb;
    const int64_t nblk = (int64_t)sort_blocks16(g, M, &pb) * items;
    if (nblk * (int64_t)g.ntiles * 6 <= ((int64_t)2 << 30)) return 1;   // hist16 + pref <= 2 GiB
  }
  return 2;
}
bool sort_uses_lds(const Geom& g) { return g.ntiles <= kMaxLdsTiles; }

// Workspace of the two-level sort, in 32-bit words of SortWork::hist: three tables of `table` words (super-tile
// counts, starts, piece starts), the level-1 histogram [workgroups][super-tiles], the level-2 one [pieces][keys]
Sort2Layout sort2_layout(const Geom& g, int64_t M) {
  const Geom g1 = coarse_geom(g);
  Sort2Layout l;
  int64_t per_block;
  l.table = g1.ntiles + 2;
  l.pieces = M / kSort2Sub + g1.ntiles;   // every super-tile: ceil(count / kSort2Sub) <= count / kSort2Sub + 1
  l.words = 3 * l.table + (int64_t)sort_blocks(g1, M, &per_block) * g1.ntiles +
            l.pieces * ((int64_t)1 << (g.sup_shift[0] + g.sup_shift[1] + g.sup_shift[2]));
  return l;
}

int sort_blocks16(const Geom& g, int64_t M, int64_t* per_block) {
  const int items = g.nitems > 1 ? g.nitems : 1;
  const int64_t m_item = M / items;
  int64_t pb = (M + 511) / 512;
  if (pb < 4096) pb = 4096;
  if (pb > 65535) pb = 65535;
  *per_block = pb;
  const int64_t bpi = (m_item + pb - 1) / pb;
  return (int)(bpi < 1 ? 1 : bpi);
}

// The short fold path of the LDS-histogram sort kernels (fold_coords<T, true>) applies
static bool quick_fold(const Geom& g, const PointsIn& in) {
  return g.fold_pow2 && in.range_mode != NUFFT_HIP_RANGE_INFINITE;
}

template <typename T, int AOS, bool FUSED, bool QF>
static hipError_t sort_lds_pass(const Geom& g, const PointsIn& in, const SortWork& w, const SortedOut<T>& out,
                                hipStream_t stream, const StageHook& hook, int nblk, int64_t per_block,
                                size_t lds) {
  hipError_t e = ensure_lds(hist_lds_kernel<T, AOS, QF>, lds);
  if (e != hipSuccess) return e;
  e = ensure_lds(scatter_lds_kernel<T, AOS, FUSED, QF>, lds);
  if (e != hipSuccess) return e;
  hook.begin(STAGE_SORT_COUNT);
  hist_lds_kernel<T, AOS, QF><<<nblk, kSortBlock, lds, stream>>>(g, in, per_block, w.hist, w.bad_count);
  hook.end(STAGE_SORT_COUNT);
  hook.begin(STAGE_SORT_SCAN);
  colscan_kernel<<<(g.ntiles + kScanCols - 1) / kScanCols, 1024, 0, stream>>>(g.ntiles, nblk, w.hist, w.tile_count);
  scan_tiles_kernel<<<1, 1024, 0, stream>>>(w.tile_count, g.ntiles, g.max_sub, g.sub_small, (int)(in.M_item / g.ntiles_item), w.tile_start, w.sub_start);
  hook.end(STAGE_SORT_SCAN);
  hook.begin(STAGE_SORT_SCATTER);
  const int staged_mode = tune_mode(g, NUFFT_HIP_TUNE_STAGED_OFF, NUFFT_HIP_TUNE_STAGED_ON);
  const bool staged = g.ntiles_item <= kStagedMaxTiles && staged_mode != 0 &&
                      (staged_mode > 0 || (in.M_item >= 4 * kStagedChunk<T> && in.M >= kSmallSortPoints));
  if (staged) {
    const size_t slds = kStagedLds<T, 1024>;
    e = ensure_lds(scatter_staged_kernel<T, AOS, FUSED, 1024, false, QF>, slds);
    if (e != hipSuccess) return e;
    scatter_staged_kernel<T, AOS, FUSED, 1024, false, QF><<<nblk, kSortBlock, slds, stream>>>(g, in, per_block, w.hist, w.tile_start, out);
  } else {
    scatter_lds_kernel<T, AOS, FUSED, QF><<<nblk, kSortBlock, lds, stream>>>(g, in, per_block, w.hist, w.tile_start, out);
  }
  hook.end(STAGE_SORT_SCATTER);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_sort(const Geom& g, const PointsIn& in_arg, const SortWork& w, const SortedOut<T>& out,
                       hipStream_t stream, const StageHook& hook) {
  hipError_t e;
  if (in_arg.M == 0) {
    e = hipMemsetAsync(w.tile_count, 0, sizeof(int32_t) * (size_t)g.ntiles, stream);
    if (e != hipSuccess) return e;
    scan_tiles_kernel<<<1, 1024, 0, stream>>>(w.tile_count, g.ntiles, g.max_sub, 0, 0, w.tile_start, w.sub_start);
    return hipGetLastError();
  }
  const int items = g.nitems > 1 ? g.nitems : 1;
  PointsIn in = in_arg;
  in.M_item = in.M / items;
  in.blocks_per_item = 1;
  const int mode = sort_mode(g, in.M);
  if (in.strengths && !(sizeof(T) == 4 && ((mode == 0 && g.rank == 2) || (mode == 1 && g.rank == 3))))
    return hipErrorInvalidValue;   // see fused_sort_supported
  if (mode == 3) {
    if constexpr (sizeof(T) == 4) {
      if (!w.tmp) return hipErrorInvalidValue;
      const Geom g1 = coarse_geom(g);
      const int key_bits = g.sup_shift[0] + g.sup_shift[1] + g.sup_shift[2];
      const int nkeys = 1 << key_bits;
      int64_t per_block;
      in.blocks_per_item = sort_blocks(g1, in.M, &per_block);
      const int nblk = in.blocks_per_item;
      const Sort2Layout lay = sort2_layout(g, in.M);
      int32_t* c_count = w.hist;
      int32_t* c_start = c_count + lay.table;
      int32_t* c_sub = c_start + lay.table;
      int32_t* hist1 = c_sub + lay.table;
      int32_t* hist2 = hist1 + (int64_t)nblk * g1.ntiles;
      uint4* tmp_rec = reinterpret_cast<uint4*>(w.tmp);
      const int aos = in.aos == 3 ? 3 : 0;
      const size_t lds1 = sizeof(int) * (size_t)g1.ntiles;
      hook.begin(STAGE_SORT_COUNT);
      const bool qf = quick_fold(g1, in);
      if (aos == 3) {
        if (qf) hist_lds_kernel<T, 3, true><<<nblk, kSortBlock, lds1, stream>>>(g1, in, per_block, hist1, w.bad_count);
        else hist_lds_kernel<T, 3, false><<<nblk, kSortBlock, lds1, stream>>>(g1, in, per_block, hist1, w.bad_count);
      } else {
        if (qf) hist_lds_kernel<T, 0, true><<<nblk, kSortBlock, lds1, stream>>>(g1, in, per_block, hist1, w.bad_count);
        else hist_lds_kernel<T, 0, false><<<nblk, kSortBlock, lds1, stream>>>(g1, in, per_block, hist1, w.bad_count);
      }
      hook.end(STAGE_SORT_COUNT);
      hook.begin(STAGE_SORT_SCAN);
      colscan_kernel<<<(g1.ntiles + kScanCols - 1) / kScanCols, 1024, 0, stream>>>(g1.ntiles, nblk, hist1, c_count);
      scan_tiles_kernel<<<1, 1024, 0, stream>>>(c_count, g1.ntiles, kSort2Sub, 0, 0, c_start, c_sub);
      hook.end(STAGE_SORT_SCAN);
      hook.begin(STAGE_SORT_SCATTER);
      {
        SortedOut<T> l1;
        l1.rec = reinterpret_cast<Rec<T>*>(tmp_rec);
        // (r06: up to 4096 super-tiles -- fine grids up to 1024^3 -- on the staged kernel's 4096-destination form: 6144 records
        // per pass, 1.5 per destination, so the level-1 writes are single records again; what is left of the gain is level 2's)
        const bool big1 = g1.ntiles > 1024;
        const size_t slds = big1 ? kStagedLds<T, 4096> : kStagedLds<T, 1024>;
#define NUFFT_SORT2_LEVEL1(AOSV, QFV)                                                                                   \
  do {                                                                                                                  \
    if (big1) {                                                                                                         \
      e = ensure_lds(scatter_staged_kernel<T, AOSV, false, 4096, true, QFV>, slds);                                     \
      if (e != hipSuccess) return e;                                                                                    \
      scatter_staged_kernel<T, AOSV, false, 4096, true, QFV><<<nblk, kSortBlock, slds, stream>>>(g1, in, per_block, hist1, \
                                                                                                 c_start, l1);         \
    } else {                                                                                                            \
      e = ensure_lds(scatter_staged_kernel<T, AOSV, false, 1024, true, QFV>, slds);                                     \
      if (e != hipSuccess) return e;                                                                                    \
      scatter_staged_kernel<T, AOSV, false, 1024, true, QFV><<<nblk, kSortBlock, slds, stream>>>(g1, in, per_block, hist1, \
                                                                                                 c_start, l1);         \
    }                                                                                                                   \
  } while (0)
        if (aos == 3) {
          if (qf) NUFFT_SORT2_LEVEL1(3, true); else NUFFT_SORT2_LEVEL1(3, false);
        } else {
          if (qf) NUFFT_SORT2_LEVEL1(0, true); else NUFFT_SORT2_LEVEL1(0, false);
        }
#undef NUFFT_SORT2_LEVEL1
      }
      const int npiece = (int)lay.pieces;
      count2_kernel<<<npiece, kSort2Threads, 0, stream>>>(g, g1, c_start, c_sub, tmp_rec, hist2, nkeys);
      scan2_kernel<<<g1.ntiles, 1024, 0, stream>>>(c_sub, hist2, nkeys, key_bits, w.tile_count);
      scan_tiles_kernel<<<1, 1024, 0, stream>>>(w.tile_count, g.ntiles, g.max_sub, g.sub_small, (int)(in.M_item / g.ntiles_item), w.tile_start, w.sub_start);
      e = ensure_lds(scatter2_kernel, kSort2Lds);
      if (e != hipSuccess) return e;
      scatter2_kernel<<<npiece, kSortBlock, kSort2Lds, stream>>>(g, g1, c_start, c_sub, tmp_rec, hist2, nkeys, key_bits, w.tile_start,
                                                                 reinterpret_cast<uint4*>(out.rec));
      hook.end(STAGE_SORT_SCATTER);
      return hipGetLastError();
    } else {
      return hipErrorInvalidValue;
    }
  }
  if (mode == 0) {
    int64_t per_block;
    in.blocks_per_item = sort_blocks(g, in.M, &per_block);
    const int nblk = in.blocks_per_item * items;
    const size_t lds = sizeof(int) * (size_t)g.ntiles;
    hipError_t (*run)(const Geom&, const PointsIn&, const SortWork&, const SortedOut<T>&, hipStream_t,
                      const StageHook&, int, int64_t, size_t) = nullptr;
    // interleaved [M, rank] points (what the op hands over): vector loads
    const int aos = (in.aos == g.rank && (g.rank == 2 || g.rank == 3)) ? g.rank : 0;
    if (quick_fold(g, in)) {
      if (in.strengths) {
        if constexpr (sizeof(T) == 4) run = aos == 2 ? sort_lds_pass<T, 2, true, true> : sort_lds_pass<T, 0, true, true>;
      } else {
        run = aos == 2 ? sort_lds_pass<T, 2, false, true> : aos == 3 ? sort_lds_pass<T, 3, false, true> : sort_lds_pass<T, 0, false, true>;
      }
    } else if (in.strengths) {
      if constexpr (sizeof(T) == 4) run = aos == 2 ? sort_lds_pass<T, 2, true, false> : sort_lds_pass<T, 0, true, false>;
    } else {
      run = aos == 2 ? sort_lds_pass<T, 2, false, false> : aos == 3 ? sort_lds_pass<T, 3, false, false> : sort_lds_pass<T, 0, false, false>;
    }
    if (!run) return hipErrorInvalidValue;
    return run(g, in, w, out, stream, hook, nblk, per_block, lds);
  }
  if (mode == 1) {
    int64_t per_block;
    in.blocks_per_item = sort_blocks16(g, in.M, &per_block);
    const int nblk = in.blocks_per_item * items;
    const int nw = (g.ntiles + 1) >> 1;
    const int ranges = (g.ntiles + kMaxLds16Tiles - 1) / kMaxLds16Tiles;
    const int span = (((g.ntiles + ranges - 1) / ranges) + 1) & ~1;         // tiles per range, even
    const size_t lds = sizeof(unsigned) * (size_t)(span >> 1);
    uint32_t* hist16 = reinterpret_cast<uint32_t*>(w.hist);                 // [nblk][nw] words
    int32_t* pref = w.hist + (int64_t)nblk * nw;                            // [nblk][ntiles]
    e = ensure_lds(hist16_lds_kernel<T>, lds);
    if (e != hipSuccess) return e;
    uint16_t* rank16 = reinterpret_cast<uint16_t*>(w.rank_of);
    hook.begin(STAGE_SORT_COUNT);
    hist16_lds_kernel<T><<<dim3(nblk, ranges), kSortBlock, lds, stream>>>(g, in, per_block, span, hist16,
                                                                          rank16, w.bad_count);
    hook.end(STAGE_SORT_COUNT);
    hook.begin(STAGE_SORT_SCAN);
    colscan16_kernel<<<(g.ntiles + 63) / 64, 1024, 0, stream>>>(
        g.ntiles, nblk, reinterpret_cast<const uint16_t*>(hist16), pref, w.tile_count);
    scan_tiles_kernel<<<1, 1024, 0, stream>>>(w.tile_count, g.ntiles, g.max_sub, g.sub_small, (int)(in.M_item / g.ntiles_item), w.tile_start, w.sub_start);
    hook.end(STAGE_SORT_SCAN);
    hook.begin(STAGE_SORT_SCATTER);
    if constexpr (sizeof(T) == 4) {
      if (in.strengths)
        scatter_ranked_kernel<T, true><<<dim3(blocks_for(in.M_item, 256), items), 256, 0, stream>>>(g, in, per_block,
                                                                                  rank16, pref, w.tile_start, out);
      else
        scatter_ranked_kernel<T><<<dim3(blocks_for(in.M_item, 256), items), 256, 0, stream>>>(g, in, per_block, rank16,
                                                                            pref, w.tile_start, out);
    } else {
      scatter_ranked_kernel<T><<<dim3(blocks_for(in.M_item, 256), items), 256, 0, stream>>>(g, in, per_block, rank16,
                                                                          pref, w.tile_start, out);
    }
    hook.end(STAGE_SORT_SCATTER);
    return hipGetLastError();
  }
  e = hipMemsetAsync(w.tile_count, 0, sizeof(int32_t) * (size_t)g.ntiles, stream);
  if (e != hipSuccess) return e;
  hook.begin(STAGE_SORT_COUNT);
  in.blocks_per_item = 1;
  count_global_kernel<T><<<dim3(blocks_for(in.M_item, 256), items), 256, 0, stream>>>(g, in, w.tile_of, w.rank_of,
                                                                    w.tile_count, w.bad_count);
  hook.end(STAGE_SORT_COUNT);
  hook.begin(STAGE_SORT_SCAN);
  scan_tiles_kernel<<<1, 1024, 0, stream>>>(w.tile_count, g.ntiles, g.max_sub, g.sub_small, (int)(in.M_item / g.ntiles_item), w.tile_start, w.sub_start);
  hook.end(STAGE_SORT_SCAN);
  hook.begin(STAGE_SORT_SCATTER);
  scatter_global_kernel<T><<<dim3(blocks_for(in.M_item, 256), items), 256, 0, stream>>>(g, in, w.tile_of, w.rank_of,
                                                                      w.tile_start, out);
  hook.end(STAGE_SORT_SCATTER);
  return hipGetLastError();
}
template hipError_t launch_sort<float>(const Geom&, const PointsIn&, const SortWork&,
                                       const SortedOut<float>&, hipStream_t, const StageHook&);
template hipError_t launch_sort<double>(const Geom&, const PointsIn&, const SortWork&,
                                        const SortedOut<double>&, hipStream_t, const StageHook&);

// Cell-grouped kernel: by point density (the in-LDS sort only pays when start cells are shared
// often enough); options.tuning GROUP_OFF / GROUP_ON force the choice.
static bool wave8_use_group(const Geom& g, int64_t M) {
  const int mode = tune_mode(g, NUFFT_HIP_TUNE_GROUP_OFF, NUFFT_HIP_TUNE_GROUP_ON);
  if (g.max_sub > kGroupMaxSub) return false;
  if (mode >= 0) return mode != 0;
  return (double)M >= kGroupMinDensity * (double)g.nf[0] * (double)g.nf[1];
}

// Defaults from the r01 sweeps (tools/sweep_w8.py, tools/sweep_w8_group.py): 4 x 64
// for the per-point kernel, 12 x 64 for the cell-grouped one (79 KB of LDS: two
// workgroups = 24 waves per CU, 3 per SIMD each; 8 x 64 = 16 waves per CU was 8 % slower).
static int wave8_nw(bool grouped) { return grouped ? NUFFT_GROUP_NW : 4; }
static int wave8_ch(bool) { return 64; }
constexpr int kW2NW = 4, kW2CH = 64;   // launch shape of spread_wave2_kernel (others measured no better)
static size_t group_lds(int nw, int ch, bool presorted, int precision = NUFFT_HIP_F32) {
  const int stage = precision == NUFFT_HIP_F32 ? kGroupStage : kGroupStage / 2;   // kGroupStageOf<T>
  return sizeof(double) * 2 * kWPlane + (size_t)precision * nw * 3 * ((ch < stage ? ch : stage) / 4) * kGroupBlk +
         (presorted ? 0 : 1024 * 4 + kGroupMaxSub * 2 + 64);   // + counters, permutation, wave sums
}
static size_t wave8_lds(bool grouped, bool presorted = false) {
  const int nw = wave8_nw(grouped), ch = wave8_ch(grouped);
  if (!grouped) return sizeof(double) * 2 * kWPlane + sizeof(float) * nw * ch * kWW * 3;
  return group_lds(nw, ch, presorted);
}
static size_t wave2_lds(const Geom& g, int precision) {
  size_t cells = (size_t)g.lstride;
  for (int d = 1; d < g.rank; ++d) cells *= (size_t)g.ldim[d];
  return (cells * 2 + 256) * sizeof(double) + (size_t)precision * kW2NW * kW2CH * 24;
}

// Geometry the cell-grouped 2-D kernels need (either precision)
static bool group2d_geometry(const Geom& g) {
  if (g.wide) return false;
  return g.rank == 2 && g.w <= kWW && g.ncoef <= kWaveCoef && g.tile[0] == kWT && g.tile[1] == kWT &&
         g.lstride == kWS;
}
// Specialised 2-D float kernels (grouped, per-point w = 8, lazy cell sort) applicable?
static bool wave8_supported(const Geom& g, int precision) {
  return precision == NUFFT_HIP_F32 && group2d_geometry(g);
}

// Wavefront-per-point kernels need w <= 8, rank 2/3 and the tile geometry that
// makes their LDS accesses conflict free.
bool wave_method_supported(const Geom& g, int precision) {
  if (g.w > 8) return false;
  if (g.rank == 2) return g.tile[0] == 32 && g.tile[1] == 32;
  // (depth 8 at w = 8: float only, one fp64 plane per launch -- see configure())
  if (g.rank == 3)
    return g.tile[0] == 16 && g.tile[1] == 16 &&
           (g.tile[2] == 4 || (g.tile[2] == 8 && (g.w <= 6 || ((g.w == 7 || g.w == 8) && precision == NUFFT_HIP_F32))));
  return false;
}
int wave_lstride(int rank) { return rank == 2 ? 40 : 24; }
int wave3_pad(int w) {   // = kWave3Pad<w>
  const int over = 551 - 24 * (15 + w) + 1;
  return over > 64 ? ((over + 7) & ~7) : 64;
}

// 3-D float w = 8 on depth-8 tiles: one launch with BOTH fp64 planes (132 KB: one workgroup of 16 waves
// per CU) instead of one launch per component, for thin point sets. There the kernel is bound by the
// write-out of the tile + halo, and a joint write-out adds (re, im) of consecutive cells with consecutive
// lanes -- every 128-byte line of the fine grid is visited once, not once per component. Measured r02 on
// 128^3 (256^3 fine cells), whole type-1 transform, split -> joint: M = 8e5 (0.05 points per cell) 1.07 ->
// 0.72 ms, 2e6 1.13 -> 1.04, 4e6 1.57 -> 1.56, 8e6 2.67 -> 2.63, 1e7 3.25 -> 3.20, 3e7 level; M = 1e8 on
// 256^3 32.0 -> 32.2. Taken below 0.5 points per cell. options.tuning JOINT_OFF / JOINT_ON force the choice.
static bool wave3_joint_wanted(const Geom& g, int64_t M) {
  const int mode = tune_mode(g, NUFFT_HIP_TUNE_JOINT_OFF, NUFFT_HIP_TUNE_JOINT_ON);
  if (mode >= 0) return mode != 0;
  return (double)M < 0.5 * (double)g.nf[0] * (double)g.nf[1] * (double)g.nf[2];
}
// one fp64 plane, 12 waves, 32-point staging chunks (the split launches of depth-4 tiles, and the
// crowded-tile fallback of the fixed-point plans)
static size_t wave3_split_lds(const Geom& g) {
  size_t cells = (size_t)g.lstride;
  for (int d = 1; d < g.rank; ++d) cells *= (size_t)g.ldim[d];
  return cells * sizeof(double) + (size_t)wave3_pad(g.w) * sizeof(double) +
         sizeof(float) * 12 * 32 * 2 * (g.w <= 6 ? 6 : 8) + 256;
}
static size_t wave3_split8_lds(const Geom& g) {   // one plane, 12 waves, 16-point staging chunks
  size_t cells = (size_t)g.lstride;
  for (int d = 1; d < g.rank; ++d) cells *= (size_t)g.ldim[d];
  return cells * sizeof(double) + (size_t)wave3_pad(g.w) * sizeof(double) + sizeof(float) * 12 * 16 * 2 * 8 + 256;
}
static size_t wave3_joint_lds(const Geom& g) {
  size_t cells = (size_t)g.lstride;
  for (int d = 1; d < g.rank; ++d) cells *= (size_t)g.ldim[d];
  return cells * 2 * sizeof(double) + (size_t)wave3_pad(g.w) * sizeof(double) + sizeof(float) * 16 * 16 * 2 * 8 + 256;
}

// Waves per workgroup of the 3-D kernel. LDS decides how many workgroups share a CU and
// 69 VGPRs allow 7 waves per SIMD: the float fixed-point form (one 52 KB plane) runs two
// workgroups of 12 waves (3 per SIMD each; config 4: 11.8 ms against 15.6 ms with 16 waves
// in ONE workgroup; 2 x 14 does not co-reside: 4+4+3+3 waves per SIMD twice exceeds the
// register file of SIMD 0); two fp64 planes leave room for one workgroup only.
template <typename T, bool FX> static constexpr int wave3d_nw() { return sizeof(T) == 4 ? (FX ? 12 : 16) : 8; }
static int wave3d_nw_rt(int precision, bool fx) { return precision == NUFFT_HIP_F32 ? (fx ? 12 : 16) : 8; }

// Upper bound on the number of subproblems, known without reading the device:
// sum_b ceil(n_b / S) <= ntiles + M / S.
static inline unsigned subproblem_grid(const Geom& g, int64_t M) {
  return (unsigned)((int64_t)g.ntiles + M / (g.sub_small > 0 && g.sub_small < g.max_sub ? g.sub_small : g.max_sub));
}

unsigned subproblem_grid_bound(const Geom& g, int64_t M) { return subproblem_grid(g, M); }

hipError_t launch_crowded_list(const Geom& g, const int32_t* sub_start, int* fb_list, hipStream_t stream) {
  const hipError_t e = hipMemsetAsync(fb_list, 0, sizeof(int), stream);
  if (e != hipSuccess) return e;
  crowded_list_kernel<<<(unsigned)((g.ntiles + 255) / 256), 256, 0, stream>>>(g, sub_start, fb_list);
  return hipGetLastError();
}

// LDS-free spreader for sparse point sets: below a few points per thousand fine cells the
// per-tile zero-fill and write-out of the LDS kernels outweigh the per-point global atomics.
// Crossover measured r02 (profiles/r02_sparse_crossover.txt, spread stage): 3-D 512^3 w = 6
// between 7.5e-4 (LDS-free 0.80 ms vs 0.99 ms) and 2.2e-3 points per cell (1.94 vs 1.34 ms);
// 2-D 2048^2 w = 8 between 2.4e-3 (25 vs 31 us) and 7.2e-3 (47 vs 34 us).
// options.tuning SPARSE_OFF / SPARSE_ON force it off / on.
bool sparse_wanted(const Geom& g, int64_t M) {
  const int mode = tune_mode(g, NUFFT_HIP_TUNE_SPARSE_OFF, NUFFT_HIP_TUNE_SPARSE_ON);
  if (mode >= 0) return mode != 0;
  const double density = g.rank == 3 ? 1.2e-3 : 4e-3;
  return (double)M < density * (double)g.nf[0] * (double)g.nf[1] * (double)g.nf[2];
}

// Records that carry the strengths (FusedRec): 2-D float, the 32 x 32-tile geometry of the
// wavefront spreaders (grouped, per-point w = 8, per-point narrower), LDS-histogram sort, and
// a point set that does not go to the LDS-free kernel.
bool fused_sort_supported(const Geom& g, int method, int precision, int64_t M) {
  if ((g.tuning & NUFFT_HIP_TUNE_NO_FUSED) || method != NUFFT_HIP_METHOD_TILE_WAVE || M <= 0) return false;
  // 3-D float fixed-point plans (nufft_dense3.hip) on the ranked-scatter sort path: 32-byte FusedRec3 records
  if (dense3_supported(g, precision))
    return g.nitems <= 1 && sort_mode(g, M) == 1 && !(g.sparse_auto && sparse_wanted(g, M));
  if (!wave8_supported(g, precision)) return false;
  if (M <= 0 || sort_mode(g, M * (g.nitems > 1 ? g.nitems : 1)) != 0 || g.max_sub > kGroupMaxSub) return false;   // M: per set
  return !(g.sparse_auto && sparse_wanted(g, M));   // every LDS-tile 2-D float spreader reads fused records
}

// By point density; options.tuning CELLSORT_OFF never, CELLSORT_ON whenever the geometry allows.
bool cellsort_wanted(const Geom& g, int method, int precision, int64_t M) {
  const int mode = tune_mode(g, NUFFT_HIP_TUNE_CELLSORT_OFF, NUFFT_HIP_TUNE_CELLSORT_ON);
  // only the 2-D w = 8 float spreader has a kernel that exploits the order
  if (method != NUFFT_HIP_METHOD_TILE_WAVE || !wave8_supported(g, precision)) return false;
  if (g.max_sub > kCellSortMaxSub || M == 0 || mode == 0) return false;
  if (mode > 0) return true;
  return wave8_use_group(g, M);
}

// The same ordering for the 3-D interp kernel (type 2 / interp op), applied eagerly in
// set_points: it costs ~1 ms per 1e8 points and removes the LDS bank conflicts of the
// stencil loop. options.tuning CELLSORT3D_OFF / CELLSORT3D_ON force it off / on.
bool cellsort_wanted_interp(const Geom& g, int method, int precision, int64_t M) {
  (void)precision;
  const int mode = tune_mode(g, NUFFT_HIP_TUNE_CELLSORT3D_OFF, NUFFT_HIP_TUNE_CELLSORT3D_ON);
  if (method != NUFFT_HIP_METHOD_TILE_WAVE || g.wide || g.rank != 3 || g.tile[0] != 16 || g.tile[1] != 16 || g.tile[2] > 8) return false;
  if (g.max_sub > kCellSortMaxSub || M == 0 || mode == 0) return false;
  if (mode > 0) return true;
  return (double)M >= kInterpSortMinDensity * (double)g.nf[0] * (double)g.nf[1] * (double)g.nf[2];
}

template <typename T>
hipError_t launch_cellsort(const Geom& g, int64_t M, const int32_t* tile_start, const int32_t* sub_start,
                           const Rec<T>* in, Rec<T>* out, hipStream_t stream) {
  if (M == 0) return hipSuccess;
  if (g.rank == 3)
    cellsort3d_kernel<T><<<subproblem_grid(g, M), kCellSortThreads, 0, stream>>>(g, tile_start, sub_start, in, out);
  else
    cellsort2d_kernel<T><<<subproblem_grid(g, M), kCellSortThreads, 0, stream>>>(g, tile_start, sub_start, in, out);
  return hipGetLastError();
}
template hipError_t launch_cellsort<float>(const Geom&, int64_t, const int32_t*, const int32_t*,
                                           const Rec<float>*, Rec<float>*, hipStream_t);
template hipError_t launch_cellsort<double>(const Geom&, int64_t, const int32_t*, const int32_t*,
                                            const Rec<double>*, Rec<double>*, hipStream_t);

size_t spread_lds_bytes(const Geom& g, int method, int precision) {
  // LDS tiles are double for both precisions
  size_t cells = (size_t)g.lstride;
  for (int d = 1; d < g.rank; ++d) cells *= (size_t)g.ldim[d];
  if (method == NUFFT_HIP_METHOD_TILE_WAVE && g.wide) return wide_spread_lds_bytes(g.rank, g.w, precision);
  if (method == NUFFT_HIP_METHOD_TILE_WAVE) {
    if (wave8_supported(g, precision))   // maximum over the kernels launch_spread may pick
      return std::max(std::max(wave8_lds(true), group_lds(12, 64, false)),
                      g.w == kWW ? wave8_lds(false) : wave2_lds(g, precision));
    if (g.rank == 2)
      return group2d_geometry(g) ? std::max(wave2_lds(g, precision), group_lds(8, 64, false, precision)) : wave2_lds(g, precision);
    if (dense3_supported(g, precision))   // (and the fp64-plane launches behind it for crowded tiles)
      return std::max(dense3_lds_bytes(g.w), wave3_split_lds(g));
    if (patch3_supported(g, precision)) return std::max(patch3_lds_bytes(g.w), wave3_split8_lds(g));
    const int nw = g.split_reim ? 12 : wave3d_nw_rt(precision, g.fixed_point != 0);
    const int ch = (g.split_reim && g.tile[2] == 8) ? 16 : 32;   // (staging chunk: keeps two workgroups per CU)
    if (g.split_reim && g.tile[2] == 8 && g.w == 8) return wave3_joint_lds(g);   // the larger of the two forms
    return cells * ((g.fixed_point || g.split_reim) ? 1 : 2) * sizeof(double) + (size_t)wave3_pad(g.w) * sizeof(double) +
           (size_t)precision * nw * ch * 2 * (g.w <= 6 ? 6 : 8) + 256;
  }
  return cells * 2 * sizeof(double);
}

size_t interp_lds_bytes(const Geom& g, int method, int precision) {
  if (g.line && method == NUFFT_HIP_METHOD_TILE_GENERIC) return line_interp_lds_bytes(g, precision);
  if (method != NUFFT_HIP_METHOD_TILE_WAVE) return 0;
  if (g.wide) return wide_interp_lds_bytes(g.rank, g.w, precision);
  size_t cells = (size_t)g.lstride;
  for (int d = 1; d < g.rank; ++d) cells *= (size_t)g.ldim[d];
  return cells * 2 * (size_t)precision;
}


template <typename T>
hipError_t launch_spread(const Geom& g, int method, const SortedPoints<T>& sp, int64_t M,
                         const T* horner, const T* c, T* fw, int batch, int64_t c_stride,
                         int64_t fw_stride, T scale, size_t lds_bytes, hipStream_t stream) {
  if (M == 0) return hipSuccess;
  dim3 grid(subproblem_grid(g, M), (unsigned)batch);
  hipError_t e = hipSuccess;
  const int64_t Md = M / (g.nitems > 1 ? g.nitems : 1);   // points per set: what the density rules look at
  if (method == NUFFT_HIP_METHOD_POINT_GLOBAL || (g.sparse_auto && !g.fused && sparse_wanted(g, Md))) {
    switch (g.rank) {
      case 1: spread_sparse_kernel<T, 1><<<grid, kBlock, 0, stream>>>(g, sp, horner, c, fw, c_stride, fw_stride, scale); break;
      case 2: spread_sparse_kernel<T, 2><<<grid, kBlock, 0, stream>>>(g, sp, horner, c, fw, c_stride, fw_stride, scale); break;
      default: spread_sparse_kernel<T, 3><<<grid, kBlock, 0, stream>>>(g, sp, horner, c, fw, c_stride, fw_stride, scale); break;
    }
    return hipGetLastError();
  }
  if (method == NUFFT_HIP_METHOD_TILE_WAVE && g.wide)
    return launch_spread_wide<T>(g, sp, M, horner, c, fw, batch, c_stride, fw_stride, scale, stream);
  // 2-D type-2 plans on 64 x 64 tiles (the interp kernel's geometry): nufft_hip_spread on such a
  // spread_only plan takes the thread-per-point tile kernel, which works on any tile; the wavefront
  // spreaders are laid out for 32 x 32 tiles and row stride 40 only.
  if (method == NUFFT_HIP_METHOD_TILE_WAVE && g.rank == 2 && (g.tile[0] != kWT || g.tile[1] != kWT || g.lstride != kWS)) {
    method = NUFFT_HIP_METHOD_TILE_GENERIC;
    lds_bytes = spread_lds_bytes(g, method, (int)sizeof(T));
  }
  if (method == NUFFT_HIP_METHOD_TILE_WAVE) {
    if constexpr (sizeof(T) == 4) {
      // 3-D fixed-point plans: the largest and the mean strength of every slot first (one streaming pass)
      if (sp.cstats && g.rank == 3 && g.fixed_point) {
        e = launch_cstats(c, c_stride, batch * (g.nitems > 1 ? g.nitems : 1), sp.cstats_blocks, sp.cstats_slots, c_stride, sp.cstats, stream);
        if (e != hipSuccess) return e;
      }
    }
    if constexpr (sizeof(T) == 4) {
      if (wave8_supported(g, 4)) {
        const bool grouped = g.cell_sorted || wave8_use_group(g, Md);
        const int shape = wave8_nw(grouped) * 100 + wave8_ch(grouped);
        lds_bytes = wave8_lds(grouped, g.cell_sorted);   // the plan's figure is the maximum over the variants
        if (grouped) {
#define NUFFT_LAUNCH_W8G(WV, NWV, CHV)                                                         \
  case NWV * 100 + CHV:                                                                        \
    if (g.fused) {                                                                             \
      e = ensure_lds(spread_2d_w8_group_kernel<float, WV, NWV, CHV, false, true>, lds_bytes);  \
      if (e != hipSuccess) return e;                                                           \
      spread_2d_w8_group_kernel<float, WV, NWV, CHV, false, true><<<grid, NWV * 64, lds_bytes, stream>>>( \
          g, sp, horner, c, fw, c_stride, fw_stride, scale);                                   \
    } else if (g.cell_sorted) {                                                                       \
      e = ensure_lds(spread_2d_w8_group_kernel<float, WV, NWV, CHV, true>, lds_bytes);                \
      if (e != hipSuccess) return e;                                                           \
      spread_2d_w8_group_kernel<float, WV, NWV, CHV, true><<<grid, NWV * 64, lds_bytes, stream>>>(    \
          g, sp, horner, c, fw, c_stride, fw_stride, scale);                                   \
    } else {                                                                                   \
      e = ensure_lds(spread_2d_w8_group_kernel<float, WV, NWV, CHV, false>, lds_bytes);               \
      if (e != hipSuccess) return e;                                                           \
      spread_2d_w8_group_kernel<float, WV, NWV, CHV, false><<<grid, NWV * 64, lds_bytes, stream>>>(   \
          g, sp, horner, c, fw, c_stride, fw_stride, scale);                                   \
    }                                                                                          \
    break;
#define NUFFT_CASE_W8G(WV)   /* narrower kernels: default shape only */                       \
  case WV:                                                                                     \
    lds_bytes = group_lds(12, 64, g.cell_sorted);                                              \
    switch (1264) {                                                                            \
      NUFFT_LAUNCH_W8G(WV, 12, 64)                                                             \
    }                                                                                          \
    break;
          switch (g.w) {
            NUFFT_CASE_W8G(2) NUFFT_CASE_W8G(3) NUFFT_CASE_W8G(4) NUFFT_CASE_W8G(5)
            NUFFT_CASE_W8G(6) NUFFT_CASE_W8G(7)
            case 8:
              switch (shape) {
                NUFFT_LAUNCH_W8G(8, NUFFT_GROUP_NW, 64)
                default: return hipErrorInvalidValue;
              }
              break;
            default: return hipErrorInvalidValue;
          }
#undef NUFFT_CASE_W8G
#undef NUFFT_LAUNCH_W8G
          return hipGetLastError();
        }
        if (g.w == kWW) {
#define NUFFT_LAUNCH_W8(NWV, CHV)                                                              \
  case NWV * 100 + CHV:                                                                        \
    if (g.fused) {                                                                             \
      e = ensure_lds(spread_2d_w8_wave_kernel<NWV, CHV, true>, lds_bytes);                     \
      if (e != hipSuccess) return e;                                                           \
      spread_2d_w8_wave_kernel<NWV, CHV, true><<<grid, NWV * 64, lds_bytes, stream>>>(         \
          g, sp, horner, c, fw, c_stride, fw_stride, scale);                                   \
    } else {                                                                                   \
      e = ensure_lds(spread_2d_w8_wave_kernel<NWV, CHV>, lds_bytes);                           \
      if (e != hipSuccess) return e;                                                           \
      spread_2d_w8_wave_kernel<NWV, CHV><<<grid, NWV * 64, lds_bytes, stream>>>(               \
          g, sp, horner, c, fw, c_stride, fw_stride, scale);                                   \
    }                                                                                          \
    break;
        switch (shape) {
          NUFFT_LAUNCH_W8(4, 64)
          default: return hipErrorInvalidValue;
        }
#undef NUFFT_LAUNCH_W8
        return hipGetLastError();
        }   // narrower kernels at low density: spread_wave2_kernel below
      }
    }
    if constexpr (sizeof(T) == 8) {
      // double precision: the same cell-grouped kernel (8 waves, 16 staged points: the byte
      // budget of the float form), in-kernel sort only
      if (group2d_geometry(g) && wave8_use_group(g, Md)) {
        lds_bytes = group_lds(8, 64, false, NUFFT_HIP_F64);
#define NUFFT_CASE_W8D(WV)                                                                     \
  case WV:                                                                                     \
    e = ensure_lds(spread_2d_w8_group_kernel<double, WV, 8, 64, false>, lds_bytes);            \
    if (e != hipSuccess) return e;                                                             \
    spread_2d_w8_group_kernel<double, WV, 8, 64, false><<<grid, 8 * 64, lds_bytes, stream>>>(  \
        g, sp, horner, c, fw, c_stride, fw_stride, scale);                                     \
    break;
        switch (g.w) {
          NUFFT_CASE_W8D(2) NUFFT_CASE_W8D(3) NUFFT_CASE_W8D(4) NUFFT_CASE_W8D(5)
          NUFFT_CASE_W8D(6) NUFFT_CASE_W8D(7) NUFFT_CASE_W8D(8)
          default: return hipErrorInvalidValue;
        }
#undef NUFFT_CASE_W8D
        return hipGetLastError();
      }
    }
    if (g.rank == 2) {
      lds_bytes = wave2_lds(g, (int)sizeof(T));
#define NUFFT_CASE_W2(WW)                                                                    \
  case WW:                                                                                   \
    if (g.fused) {                                                                           \
      if constexpr (sizeof(T) == 4) {                                                        \
        e = ensure_lds(spread_wave2_kernel<T, WW, kW2NW, kW2CH, true>, lds_bytes);           \
        if (e != hipSuccess) return e;                                                       \
        spread_wave2_kernel<T, WW, kW2NW, kW2CH, true><<<grid, kW2NW * 64, lds_bytes, stream>>>( \
            g, sp, horner, c, fw, c_stride, fw_stride, scale);                               \
      } else { return hipErrorInvalidValue; }                                                \
    } else {                                                                                 \
      e = ensure_lds(spread_wave2_kernel<T, WW, kW2NW, kW2CH>, lds_bytes);                   \
      if (e != hipSuccess) return e;                                                         \
      spread_wave2_kernel<T, WW, kW2NW, kW2CH><<<grid, kW2NW * 64, lds_bytes, stream>>>(     \
          g, sp, horner, c, fw, c_stride, fw_stride, scale);                                 \
    }                                                                                        \
    break;
      switch (g.w) {
        NUFFT_CASE_W2(2) NUFFT_CASE_W2(3) NUFFT_CASE_W2(4) NUFFT_CASE_W2(5)
        NUFFT_CASE_W2(6) NUFFT_CASE_W2(7) NUFFT_CASE_W2(8)
        default: return hipErrorInvalidValue;
      }
#undef NUFFT_CASE_W2
    } else {
#define NUFFT_LAUNCH_W3(WW, TZV, FXV)                                                            \
  e = ensure_lds(spread_wave3_kernel<T, WW, TZV, wave3d_nw<T, FXV>(), 32, FXV>, lds_bytes);       \
  if (e != hipSuccess) return e;                                                                 \
  spread_wave3_kernel<T, WW, TZV, wave3d_nw<T, FXV>(), 32, FXV>                                   \
      <<<grid, wave3d_nw<T, FXV>() * 64, lds_bytes, stream>>>(g, sp, horner, c, fw, c_stride, fw_stride, scale);
// (fgrid: a persistent grid over the fallback list when the launch runs behind a fixed-point spreader)
#define NUFFT_LAUNCH_W3S(WW, TZV, CV)                                                            \
  e = ensure_lds(spread_wave3_kernel<T, WW, TZV, 12, 32, false, CV>, lds_bytes);                  \
  if (e != hipSuccess) return e;                                                                 \
  spread_wave3_kernel<T, WW, TZV, 12, 32, false, CV>                                              \
      <<<fgrid, 12 * 64, lds_bytes, stream>>>(g, sp, horner, c, fw, c_stride, fw_stride, scale);
#define NUFFT_LAUNCH_W3S8(WW, CV)                                                                \
  e = ensure_lds(spread_wave3_kernel<T, WW, 8, 12, 16, false, CV>, lds_bytes);                    \
  if (e != hipSuccess) return e;                                                                 \
  spread_wave3_kernel<T, WW, 8, 12, 16, false, CV>                                                \
      <<<fgrid, 12 * 64, lds_bytes, stream>>>(g, sp, horner, c, fw, c_stride, fw_stride, scale);
#define NUFFT_CASE_W3(WW)                                                                        \
  case WW:                                                                                       \
    if (g.tile[2] == 8) {                                                                        \
      if constexpr (WW >= 7) {                                                                   \
        if constexpr (sizeof(T) == 4) {                                                          \
          if (g.fx_patch) {   /* packed fixed point, exact conversion; flagged subproblems on fp64 planes behind it */ \
            if (g.stack && sp.segs) e = launch_spread_stack3(g, sp, M, horner, c, fw, batch, c_stride, fw_stride, scale, stream); \
            else e = launch_spread_patch3(g, sp, grid.x, horner, c, fw, batch, c_stride, fw_stride, scale, stream); \
            if (e != hipSuccess) return e;                                                       \
            e = hipErrorNotSupported;                                                            \
            if (!(g.tuning & NUFFT_HIP_TUNE_FBGROUP_OFF)) {   /* r05: cell-grouped, both planes in one launch */ \
              e = launch_spread_group3_fallback(g, sp, grid.x, horner, c, fw, batch, c_stride, fw_stride, scale, stream); \
              if (e != hipSuccess && e != hipErrorNotSupported) return e;                        \
            }                                                                                    \
            if (e == hipErrorNotSupported) {   /* (forced off, or a part without room for its 150 KB of LDS) */ \
              lds_bytes = wave3_split8_lds(g);                                                   \
              NUFFT_LAUNCH_W3S8(WW, 1) NUFFT_LAUNCH_W3S8(WW, 2)                                   \
            }                                                                                    \
          } else if constexpr (WW == 7) { return hipErrorInvalidValue;                           \
          } else if (!g.split_reim) { return hipErrorInvalidValue;                               \
          } else if (wave3_joint_wanted(g, Md)) {   /* thin point sets: both planes, one write-out */   \
            lds_bytes = wave3_joint_lds(g);                                                      \
            e = ensure_lds(spread_wave3_kernel<T, WW, 8, 16, 16, false, 0>, lds_bytes);           \
            if (e != hipSuccess) return e;                                                       \
            spread_wave3_kernel<T, WW, 8, 16, 16, false, 0>                                       \
                <<<grid, 16 * 64, lds_bytes, stream>>>(g, sp, horner, c, fw, c_stride, fw_stride, scale); \
          } else {                                                                               \
            lds_bytes = wave3_split8_lds(g);   /* (the plan's figure is the joint form's) */     \
            NUFFT_LAUNCH_W3S8(WW, 1) NUFFT_LAUNCH_W3S8(WW, 2)                                     \
          }                                                                                      \
        } else { return hipErrorInvalidValue; }                                                  \
      } else if constexpr (WW <= 6) {                                                            \
        if (g.fixed_point) {                                                                     \
          if constexpr (sizeof(T) == 4) {                                                        \
            if (g.stack && sp.segs) e = launch_spread_dense3_stack(g, sp, M, horner, c, fw, batch, c_stride, fw_stride, scale, stream); \
            else e = launch_spread_dense3(g, sp, grid.x, Md, horner, c, fw, batch, c_stride, fw_stride, scale, stream); \
            if (e != hipSuccess) return e;                                                       \
            if (Md > (int64_t)g.fx_max_subs * g.max_sub) {   /* a tile may be crowded: fp64 planes for those */ \
              e = hipErrorNotSupported;                                                          \
              if (!(g.tuning & NUFFT_HIP_TUNE_FBGROUP_OFF) && g.ncoef <= 10) {   /* r05: cell-grouped, both planes */ \
                e = launch_spread_group3_fallback(g, sp, grid.x, horner, c, fw, batch, c_stride, fw_stride, scale, stream); \
                if (e != hipSuccess && e != hipErrorNotSupported) return e;                      \
              }                                                                                  \
              if (e == hipErrorNotSupported) {                                                   \
                lds_bytes = wave3_split_lds(g);                                                  \
                NUFFT_LAUNCH_W3S(WW, 8, 1) NUFFT_LAUNCH_W3S(WW, 8, 2)                             \
              }                                                                                  \
            }                                                                                    \
          } else { return hipErrorInvalidValue; }                                                \
        } else if (g.stack && sp.segs) {   /* r06: double precision over stacks of tiles */        \
          if constexpr (sizeof(T) == 8) {                                                        \
            const dim3 sgrid(stack_grid_bound(g, M), grid.y);                                    \
            e = ensure_lds(spread_wave3_stack_kernel<T, WW, 8, 8, 32>, lds_bytes);                \
            if (e != hipSuccess) return e;                                                       \
            spread_wave3_stack_kernel<T, WW, 8, 8, 32><<<sgrid, 8 * 64, lds_bytes, stream>>>(     \
                g, sp, horner, c, fw, c_stride, fw_stride, scale);                               \
          } else { return hipErrorInvalidValue; }                                                \
        } else { NUFFT_LAUNCH_W3(WW, 8, false) }                                                 \
      } else { return hipErrorInvalidValue; }                                                    \
    } else if (g.tile[2] == 4) {                                                                 \
      if (g.fixed_point) {                                                                       \
        if constexpr (sizeof(T) == 4) {                                                          \
          NUFFT_LAUNCH_W3(WW, 4, true)                                                           \
          if (Md > (int64_t)g.fx_max_subs * g.max_sub) {                                         \
            lds_bytes = wave3_split_lds(g);                                                      \
            NUFFT_LAUNCH_W3S(WW, 4, 1) NUFFT_LAUNCH_W3S(WW, 4, 2)                                 \
          }                                                                                      \
        } else { return hipErrorInvalidValue; }                                                  \
      } else if (g.split_reim) {                                                                 \
        if constexpr (sizeof(T) == 4) { NUFFT_LAUNCH_W3S(WW, 4, 1) NUFFT_LAUNCH_W3S(WW, 4, 2) }   \
        else { return hipErrorInvalidValue; }                                                    \
      } else if (g.stack && sp.segs) {   /* r06: double precision over stacks of tiles */          \
        if constexpr (sizeof(T) == 8) {                                                          \
          const dim3 sgrid(stack_grid_bound(g, M), grid.y);                                      \
          e = ensure_lds(spread_wave3_stack_kernel<T, WW, 4, 8, 32>, lds_bytes);                  \
          if (e != hipSuccess) return e;                                                         \
          spread_wave3_stack_kernel<T, WW, 4, 8, 32><<<sgrid, 8 * 64, lds_bytes, stream>>>(       \
              g, sp, horner, c, fw, c_stride, fw_stride, scale);                                 \
        } else { return hipErrorInvalidValue; }                                                  \
      } else { NUFFT_LAUNCH_W3(WW, 4, false) }                                                   \
    } else { return hipErrorInvalidValue; }                                                      \
    break;
      const dim3 fgrid = (g.fixed_point && sp.fb_list) ? dim3(std::min(grid.x, 2048u), grid.y) : grid;
      switch (g.w) {
        NUFFT_CASE_W3(2) NUFFT_CASE_W3(3) NUFFT_CASE_W3(4) NUFFT_CASE_W3(5)
        NUFFT_CASE_W3(6) NUFFT_CASE_W3(7) NUFFT_CASE_W3(8)
        default: return hipErrorInvalidValue;
      }
#undef NUFFT_CASE_W3
#undef NUFFT_LAUNCH_W3S8
#undef NUFFT_LAUNCH_W3S
#undef NUFFT_LAUNCH_W3
    }
    return hipGetLastError();
  }
  switch (g.rank) {
    case 1:
      e = ensure_lds(spread_tile_generic_kernel<T, 1>, lds_bytes);
      if (e != hipSuccess) return e;
      spread_tile_generic_kernel<T, 1><<<grid, kBlock, lds_bytes, stream>>>(
          g, sp, horner, c, fw, c_stride, fw_stride, scale);
      break;
    case 2:
      e = ensure_lds(spread_tile_generic_kernel<T, 2>, lds_bytes);
      if (e != hipSuccess) return e;
      spread_tile_generic_kernel<T, 2><<<grid, kBlock, lds_bytes, stream>>>(
          g, sp, horner, c, fw, c_stride, fw_stride, scale);
      break;
    default:
      e = ensure_lds(spread_tile_generic_kernel<T, 3>, lds_bytes);
      if (e != hipSuccess) return e;
      spread_tile_generic_kernel<T, 3><<<grid, kBlock, lds_bytes, stream>>>(
          g, sp, horner, c, fw, c_stride, fw_stride, scale);
      break;
  }
  return hipGetLastError();
}
template hipError_t launch_spread<float>(const Geom&, int, const SortedPoints<float>&, int64_t,
                                         const float*, const float*, float*, int, int64_t, int64_t,
                                         float, size_t, hipStream_t);
template hipError_t launch_spread<double>(const Geom&, int, const SortedPoints<double>&, int64_t,
                                          const double*, const double*, double*, int, int64_t,
                                          int64_t, double, size_t, hipStream_t);

template <typename T>
hipError_t launch_interp(const Geom& g, int method, const SortedPoints<T>& sp, int64_t M,
                         const T* horner, T* c, const T* fw, int batch, int64_t c_stride,
                         int64_t fw_stride, T scale, hipStream_t stream) {
  if (M == 0) return hipSuccess;
  dim3 grid(subproblem_grid(g, M), (unsigned)batch);
  if (g.line && method == NUFFT_HIP_METHOD_TILE_GENERIC)
    return launch_interp_line<T>(g, sp, M, horner, c, fw, batch, c_stride, fw_stride, scale, stream);
  if (method == NUFFT_HIP_METHOD_TILE_WAVE && g.wide)
    return launch_interp_wide<T>(g, sp, M, horner, c, fw, batch, c_stride, fw_stride, scale, stream);
  if (method == NUFFT_HIP_METHOD_TILE_WAVE) {
    const size_t lds = interp_lds_bytes(g, method, (int)sizeof(T));
    hipError_t e = hipSuccess;
    // 3-D: eight lanes per point while a tile holds fewer points than kSplitPointsPerTile on average (options.tuning
    // ISPLIT_OFF / ISPLIT_ON force the choice)
    bool split3 = false;
    if (g.rank == 3) {
      const int mode = tune_mode(g, NUFFT_HIP_TUNE_ISPLIT_OFF, NUFFT_HIP_TUNE_ISPLIT_ON);
      const double tiles = (double)g.ntile[0] * g.ntile[1] * g.ntile[2] * (g.nitems > 1 ? g.nitems : 1);
      split3 = mode >= 0 ? mode != 0 : (g.w >= 7 && (double)M < kSplitPointsPerTile * tiles);
    }
#define NUFFT_LAUNCH_IP(RR, WW)                                                                       \
  case RR * 100 + WW:                                                                                 \
    if constexpr (RR == 2 && sizeof(T) == 4) {                                                        \
      if (g.tile[0] == 64 && g.tile[1] == 64) {   /* big type-2 tiles: 512 threads */                 \
        e = ensure_lds(interp_point_kernel<T, RR, WW, 512>, lds);                                     \
        if (e != hipSuccess) return e;                                                                \
        interp_point_kernel<T, RR, WW, 512><<<grid, 512, lds, stream>>>(g, sp, horner, c, fw,         \
                                                                        c_stride, fw_stride, scale);  \
        break;                                                                                        \
      }                                                                                               \
    }                                                                                                 \
    if constexpr (RR == 3 && sizeof(T) == 8) {   /* (float: measured and rejected, profiles/r06_float_interp_stack_ab.txt) */ \
      if (g.stack && sp.segs && g.tile[0] == 16 && g.tile[1] == 16) {   /* r06: over stacks of tiles */ \
        const dim3 sgrid(stack_grid_bound(g, M), grid.y);                                             \
        if (split3) {                                                                                 \
          e = ensure_lds(interp_point_kernel<T, RR, WW, kInterpThreads<RR>, true, true>, lds);        \
          if (e != hipSuccess) return e;                                                              \
          interp_point_kernel<T, RR, WW, kInterpThreads<RR>, true, true><<<sgrid, kInterpThreads<RR>, lds, stream>>>( \
              g, sp, horner, c, fw, c_stride, fw_stride, scale);                                      \
        } else {                                                                                      \
          e = ensure_lds(interp_point_kernel<T, RR, WW, kInterpThreads<RR>, true>, lds);              \
          if (e != hipSuccess) return e;                                                              \
          interp_point_kernel<T, RR, WW, kInterpThreads<RR>, true><<<sgrid, kInterpThreads<RR>, lds, stream>>>( \
              g, sp, horner, c, fw, c_stride, fw_stride, scale);                                      \
        }                                                                                             \
        break;                                                                                        \
      }                                                                                               \
    }                                                                                                 \
    if constexpr (RR == 3) {                                                                          \
      if (split3) {   /* r06: eight lanes per point on tiles that hold few points */                   \
        e = ensure_lds(interp_point_kernel<T, RR, WW, kInterpThreads<RR>, false, true>, lds);         \
        if (e != hipSuccess) return e;                                                                \
        interp_point_kernel<T, RR, WW, kInterpThreads<RR>, false, true><<<grid, kInterpThreads<RR>, lds, stream>>>( \
            g, sp, horner, c, fw, c_stride, fw_stride, scale);                                        \
        break;                                                                                        \
      }                                                                                               \
    }                                                                                                 \
    e = ensure_lds(interp_point_kernel<T, RR, WW>, lds);                                              \
    if (e != hipSuccess) return e;                                                                    \
    interp_point_kernel<T, RR, WW><<<grid, kInterpThreads<RR>, lds, stream>>>(g, sp, horner, c, fw,   \
                                                                              c_stride, fw_stride, scale); \
    break;
    switch (g.rank * 100 + g.w) {
      NUFFT_LAUNCH_IP(2, 2) NUFFT_LAUNCH_IP(2, 3) NUFFT_LAUNCH_IP(2, 4) NUFFT_LAUNCH_IP(2, 5)
      NUFFT_LAUNCH_IP(2, 6) NUFFT_LAUNCH_IP(2, 7) NUFFT_LAUNCH_IP(2, 8)
      NUFFT_LAUNCH_IP(3, 2) NUFFT_LAUNCH_IP(3, 3) NUFFT_LAUNCH_IP(3, 4) NUFFT_LAUNCH_IP(3, 5)
      NUFFT_LAUNCH_IP(3, 6) NUFFT_LAUNCH_IP(3, 7) NUFFT_LAUNCH_IP(3, 8)
      default: return hipErrorInvalidValue;
    }
#undef NUFFT_LAUNCH_IP
    return hipGetLastError();
  }
  switch (g.rank) {
    case 1:
      interp_tile_generic_kernel<T, 1><<<grid, kBlock, 0, stream>>>(g, sp, horner, c, fw, c_stride,
                                                                    fw_stride, scale);
      break;
    case 2:
      interp_tile_generic_kernel<T, 2><<<grid, kBlock, 0, stream>>>(g, sp, horner, c, fw, c_stride,
                                                                    fw_stride, scale);
      break;
    default:
      interp_tile_generic_kernel<T, 3><<<grid, kBlock, 0, stream>>>(g, sp, horner, c, fw, c_stride,
                                                                    fw_stride, scale);
      break;
  }
  return hipGetLastError();
}
template hipError_t launch_interp<float>(const Geom&, int, const SortedPoints<float>&, int64_t,
                                         const float*, float*, const float*, int, int64_t, int64_t,
                                         float, hipStream_t);
template hipError_t launch_interp<double>(const Geom&, int, const SortedPoints<double>&, int64_t,
                                          const double*, double*, const double*, int, int64_t,
                                          int64_t, double, hipStream_t);

// Interpolation straight from unsorted points (interp_direct_kernel): rank 2 / 3, w <= 8, one point set.
bool direct_interp_supported(const Geom& g) { return (g.rank == 2 || g.rank == 3) && g.w <= 8 && g.nitems <= 1 && !g.wide; }
template <typename T>
hipError_t launch_interp_direct(const Geom& g, const PointsIn& in, const T* horner, T* c, const T* fw, int batch,
                                int64_t c_stride, int64_t fw_stride, T scale, hipStream_t stream) {
  if (in.M == 0) return hipSuccess;
  const dim3 grid((unsigned)((in.M + 255) / 256), (unsigned)batch);
  if (g.rank == 2) interp_direct_kernel<T, 2><<<grid, 256, 0, stream>>>(g, in, horner, c, fw, c_stride, fw_stride, scale);
  else interp_direct_kernel<T, 3><<<grid, 256, 0, stream>>>(g, in, horner, c, fw, c_stride, fw_stride, scale);
  return hipGetLastError();
}
template hipError_t launch_interp_direct<float>(const Geom&, const PointsIn&, const float*, float*, const float*, int, int64_t,
                                                int64_t, float, hipStream_t);
template hipError_t launch_interp_direct<double>(const Geom&, const PointsIn&, const double*, double*, const double*, int, int64_t,
                                                 int64_t, double, hipStream_t);

template <typename T>
hipError_t launch_deconvolve(const Geom& g, int dir, T* f, T* fw, const T* const rfser[3],
                             int batch, hipStream_t stream) {
  const int d0 = dir == 1 ? g.nmodes[0] : g.nf[0];
  const int d1 = dir == 1 ? g.nmodes[1] : g.nf[1];
  const int d2 = dir == 1 ? g.nmodes[2] : g.nf[2];
  if (d0 == 0 || d1 == 0 || d2 == 0 || batch == 0) return hipSuccess;
  const int64_t nel = (int64_t)g.nmodes[0] * g.nmodes[1] * g.nmodes[2];
  const int64_t nfel = (int64_t)g.nf[0] * g.nf[1] * g.nf[2];
  if (d1 <= 65535 && d2 <= 65535) {
    // gridDim.z = d2 * batch is limited to 65535: several launches for large batches
    const int per_launch = std::max(1, 65535 / d2);
    for (int b0 = 0; b0 < batch; b0 += per_launch) {
      const int nb = std::min(per_launch, batch - b0);
      dim3 grid(blocks_for(d0, 256), (unsigned)d1, (unsigned)(d2 * nb));
      deconvolve_kernel<T, false><<<grid, 256, 0, stream>>>(g, dir, nb, f + 2 * b0 * nel, fw + 2 * b0 * nfel,
                                                            rfser[0], rfser[1], rfser[2]);
    }
    return hipGetLastError();
  }
  // more than 65535 rows in a dimension (e.g. a 40000 x 40000 type-2 fine grid): the rows
  // (i1, i2) of one transform are flattened over gridDim.y x gridDim.z, one launch per transform
  for (int b0 = 0; b0 < batch; ++b0) {
    const int64_t rows = (int64_t)d1 * d2;
    const unsigned gy = (unsigned)std::min<int64_t>(rows, 32768);
    const unsigned gz = (unsigned)((rows + gy - 1) / gy);
    if (gz > 65535) return hipErrorInvalidValue;   // > 2^31 rows: beyond the 2e9-element cap anyway
    dim3 grid(blocks_for(d0, 256), gy, gz);
    deconvolve_kernel<T, true><<<grid, 256, 0, stream>>>(g, dir, 1, f + 2 * b0 * nel, fw + 2 * b0 * nfel,
                                                         rfser[0], rfser[1], rfser[2]);
  }
  return hipGetLastError();
}
template hipError_t launch_deconvolve<float>(const Geom&, int, float*, float*, const float* const[3],
                                             int, hipStream_t);
template hipError_t launch_deconvolve<double>(const Geom&, int, double*, double*,
                                              const double* const[3], int, hipStream_t);

hipError_t launch_permute(const void* src, void* dst, int elem_bytes, int ndim,
                          const int64_t* out_shape, const int64_t* src_strides,
                          hipStream_t stream) {
  PermuteArgs a;
  a.ndim = ndim;
  a.total = 1;
  for (int d = 0; d < ndim; ++d) {
    a.shape[d] = out_shape[d];
    a.stride[d] = src_strides[d];
    a.total *= out_shape[d];
  }
  if (a.total == 0) return hipSuccess;
  const unsigned nb = blocks_for(a.total, 256);
  switch (elem_bytes) {
    case 4:
      permute_kernel<float><<<nb, 256, 0, stream>>>((const float*)src, (float*)dst, a);
      break;
    case 8:
      permute_kernel<double><<<nb, 256, 0, stream>>>((const double*)src, (double*)dst, a);
      break;
    case 16:
      permute_kernel<double2><<<nb, 256, 0, stream>>>((const double2*)src, (double2*)dst, a);
      break;
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace nufft_hip
