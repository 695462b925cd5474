// Device-side helpers shared by the kernel translation units (nufft_kernels.hip,
// nufft_wide.hip): LDS / global accumulation, sorted-record decoding, tile row walks,
// the subproblem lookup. Everything is inline and file-local (anonymous namespace).
#pragma once
#include <hip/hip_runtime.h>
#include "nufft_hip_internal.h"

namespace nufft_hip {
namespace {

__device__ __forceinline__ void lds_add(double* p, double v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void glb_add(float* p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void glb_add(double* p, double v) { unsafeAtomicAdd(p, v); }

// Decoded view of a sorted record.
template <typename T>
struct PointView { uint32_t loc; T z0, z1, z2; int idx; };

template <typename T, int RANK>
__device__ __forceinline__ PointView<T> unpack_rec(const Rec<T>& r);
template <>
__device__ __forceinline__ PointView<float> unpack_rec<float, 1>(const Rec<float>& r) {
  return {r.loc, r.z0, r.z1, 0.f, r.idx};
}
template <>
__device__ __forceinline__ PointView<float> unpack_rec<float, 2>(const Rec<float>& r) {
  return {r.loc, r.z0, r.z1, 0.f, r.idx};
}
template <>
__device__ __forceinline__ PointView<float> unpack_rec<float, 3>(const Rec<float>& r) {
  const uint32_t w0 = r.loc, w1 = __float_as_uint(r.z0), w2 = __float_as_uint(r.z1);
  PointView<float> v;
  v.loc = (w0 >> 28) | ((w1 >> 28) << 10) | ((w2 >> 28) << 20);
  v.z0 = (float)(w0 & 0x0fffffffu) * 7.450580596923828e-09f - 1.0f;   // 2^-27
  v.z1 = (float)(w1 & 0x0fffffffu) * 7.450580596923828e-09f - 1.0f;
  v.z2 = (float)(w2 & 0x0fffffffu) * 7.450580596923828e-09f - 1.0f;
  v.idx = r.idx;
  return v;
}
template <> __device__ __forceinline__ PointView<double> unpack_rec<double, 1>(const Rec<double>& r) { return {r.loc, r.z0, r.z1, r.z2, r.idx}; }
template <> __device__ __forceinline__ PointView<double> unpack_rec<double, 2>(const Rec<double>& r) { return {r.loc, r.z0, r.z1, r.z2, r.idx}; }
template <> __device__ __forceinline__ PointView<double> unpack_rec<double, 3>(const Rec<double>& r) { return {r.loc, r.z0, r.z1, r.z2, r.idx}; }

// One (re, im) cell of an LDS tile. float: as ONE ds_read_b64 -- left to itself the compiler pairs the reads of
// neighbouring cells into ds_read2_b64, which the LDS serves at half the rate (128 instead of 256 bytes per clock,
// 16-lane groups on 32 banks instead of 32-lane groups on 64: MI355X_MICROARCH.md, LDS); a volatile access is not paired.
__device__ __forceinline__ float2 lds_cell(const float2* p) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  typedef const volatile __attribute__((address_space(3))) v2f* lds_ptr;   // (explicitly LDS: a volatile generic access is a flat load)
  const v2f v = *(lds_ptr)(p);
  return make_float2(v.x, v.y);
}
__device__ __forceinline__ double2 lds_cell(const double2* p) { return *p; }

// Tile id <-> tile coordinates (Geom::sup_shift)
__device__ __forceinline__ int tile_id(const Geom& g, const int tc[3]) {
  const int sh0 = g.sup_shift[0], sh1 = g.sup_shift[1], sh2 = g.sup_shift[2];
  const int s = (tc[0] >> sh0) + g.nsup[0] * ((tc[1] >> sh1) + g.nsup[1] * (tc[2] >> sh2));
  const int key = (tc[0] & ((1 << sh0) - 1)) | ((tc[1] & ((1 << sh1) - 1)) << sh0) | ((tc[2] & ((1 << sh2) - 1)) << (sh0 + sh1));
  return (s << (sh0 + sh1 + sh2)) | key;
}
__device__ __forceinline__ void tile_coords(const Geom& g, int tb, int* t0, int* t1, int* t2) {
  const int sh0 = g.sup_shift[0], sh1 = g.sup_shift[1], sh2 = g.sup_shift[2];
  const int key = tb & ((1 << (sh0 + sh1 + sh2)) - 1);
  const int s = tb >> (sh0 + sh1 + sh2);
  const int s0 = s % g.nsup[0], s1 = (s / g.nsup[0]) % g.nsup[1], s2 = s / (g.nsup[0] * g.nsup[1]);
  *t0 = (s0 << sh0) | (key & ((1 << sh0) - 1));
  *t1 = (s1 << sh1) | ((key >> sh0) & ((1 << sh1) - 1));
  *t2 = (s2 << sh2) | (key >> (sh0 + sh1));
}

// Record j of an array whose records lie `stride` bytes apart (16 / 32 for float, see FusedRec3)
template <typename T>
__device__ __forceinline__ const Rec<T>& rec_at(const Rec<T>* base, int j, int stride) {
  return *reinterpret_cast<const Rec<T>*>(reinterpret_cast<const unsigned char*>(base) + (size_t)j * (size_t)stride);
}

// Rows (a1, a2) of an LDS tile are dealt to waves; lanes run along x. No
// integer division or 64-bit modulo per cell (a generic `i % L0`, `% nf` walk
// cost ~20 us per 16x16x4 tile, i.e. 10 ms of the 3-D spread at 131072 tiles):
// the row counters advance incrementally and periodic wrap is one conditional
// subtract (o + a < 2 nf always, since tile <= nf and w <= nf / 2).
struct RowWalk {
  int a1, a2;
  __device__ __forceinline__ RowWalk(int first, int L1) : a1(first), a2(0) {
    while (a1 >= L1) { a1 -= L1; ++a2; }
  }
  __device__ __forceinline__ void advance(int step, int L1) {
    a1 += step;
    while (a1 >= L1) { a1 -= L1; ++a2; }
  }
};
// o + a < 3 n always (tile <= n, w <= n / 2), and < 2 n except for tiny grids
// such as n = 18 with 16-wide tiles: two conditional subtracts cover every case.
__device__ __forceinline__ int wrap1(int v, int n) {
  v = v >= n ? v - n : v;
  return v >= n ? v - n : v;
}

// Adds a finished LDS tile (planar double re/im) to the periodic fine grid.
// Consecutive lanes take (re, im) of consecutive cells of one row, so a
// global_atomic_add_f32 wave-instruction covers contiguous bytes.
template <typename T, int RANK>
__device__ __forceinline__ void tile_to_grid(const Geom& g, const double* plane_re, const double* plane_im,
                                             int LS, int PS, int tb, T* __restrict__ out, int wave,
                                             int nwaves, int lane) {
  const int L0 = g.ldim[0], L1 = g.ldim[1];
  const int L2 = RANK > 2 ? g.ldim[2] : 1;
  int t0, t1, t2;
  tile_coords(g, tb, &t0, &t1, &t2);
  const int o0 = t0 * g.tile[0], o1 = t1 * g.tile[1], o2 = t2 * g.tile[2];
  for (RowWalk r(wave, L1); r.a2 < L2; r.advance(nwaves, L1)) {
    const int g1 = wrap1(o1 + r.a1, g.nf[1]);
    const int g2 = RANK > 2 ? wrap1(o2 + r.a2, g.nf[2]) : 0;
    const int64_t rowbase = (int64_t)g.nf[0] * (g1 + (int64_t)g.nf[1] * g2);
    const int lrow = r.a2 * PS + r.a1 * LS;
    for (int e = lane; e < 2 * L0; e += 64) {
      const int a0 = e >> 1, comp = e & 1;
      const T v = (T)(comp ? plane_im : plane_re)[lrow + a0];
      if (v != (T)0) glb_add(&out[2 * (rowbase + wrap1(o0 + a0, g.nf[0])) + comp], v);
    }
  }
}

// Which subproblem does workgroup `s` own? sub_start is the exclusive scan of
// per-tile subproblem counts; returns the tile and the point range.
// With several point sets in one plan (Geom::nitems > 1) the tile index is composite,
// item * ntiles_item + tile: *tile gets the tile inside its item and *slot the index of the
// (item, transform) pair this workgroup works on, item * gridDim.y + blockIdx.y -- the
// strengths of that pair start at c + slot * c_stride, its fine grid at fw + slot * fw_stride.
__device__ __forceinline__ bool locate_subproblem(const Geom& g, const int32_t* __restrict__ tile_start,
                                                  const int32_t* __restrict__ sub_start, int s,
                                                  int* tile, int* p0, int* p1, int* slot, int* nsub = nullptr,
                                                  int* chunk_of = nullptr, int* tile_end = nullptr) {
  const int nt = g.ntiles;
  if (s >= sub_start[nt]) return false;
  // Invariant: sub_start[lo] <= s < sub_start[hi]. Most tiles own exactly one
  // subproblem, so the answer is near s: gallop outwards from that guess
  // before bisecting (2-4 dependent loads instead of log2(ntiles) = 12-18).
  int lo, hi;
  const int guess = s < nt ? s : nt - 1;
  if (sub_start[guess] <= s) {
    lo = guess;
    int step = 1;
    hi = lo + 1;
    while (hi < nt && sub_start[hi] <= s) { lo = hi; step <<= 1; hi = lo + step; }
    if (hi > nt) hi = nt;
  } else {
    hi = guess;
    int step = 1;
    lo = hi - 1;
    while (lo > 0 && sub_start[lo] > s) { hi = lo; step <<= 1; lo = hi - step; }
    if (lo < 0) lo = 0;
  }
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (sub_start[mid] <= s) lo = mid; else hi = mid;
  }
  // the tile's k subproblems split its points EVENLY (ceil(n / k) each, the last one the rest): a
  // tile just above the cap becomes two halves rather than a full and a nearly empty workgroup
  const int chunk = s - sub_start[lo];
  const int b = tile_start[lo];
  const int e = tile_start[lo + 1];
  const int k = sub_start[lo + 1] - sub_start[lo];
  const int sz = (e - b + k - 1) / k;
  const int a = b + chunk * sz;
  if (nsub) *nsub = k;   // subproblems of this tile
  if (chunk_of) *chunk_of = chunk;   // which of them this is, and where the tile's points end
  if (tile_end) *tile_end = e;
  if (g.nitems > 1) {
    const int item = lo / g.ntiles_item;
    *tile = lo - item * g.ntiles_item;
    *slot = item * (int)gridDim.y + (int)blockIdx.y;
  } else {
    *tile = lo;
    *slot = (int)blockIdx.y;
  }
  *p0 = a;
  *p1 = (a + sz < e) ? a + sz : e;
  return true;
}

__device__ __forceinline__ float bcast_lane(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ double bcast_lane(double v, int lane) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// Exclusive scan of NK LDS counters by NT threads (NK a multiple of the scanning thread count); wsum: [16] scratch.
// Ends with a barrier.
template <int NT, int NK>
__device__ __forceinline__ void scan_counts(uint32_t* cnt, uint32_t* wsum, int tid) {
  constexpr int NS = NT >= 1024 ? 1024 : NT >= 512 ? 512 : NT >= 256 ? 256 : NT >= 128 ? 128 : 64;
  constexpr int PER = NK / NS;
  const int lane = tid & 63, wave = tid >> 6;
  const bool on = tid < NS;
  uint32_t v[PER], tot = 0u;
#pragma unroll
  for (int u = 0; u < PER; ++u) { v[u] = on ? cnt[tid * PER + u] : 0u; tot += v[u]; }
  uint32_t incl = tot;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  if (on && lane == 63) wsum[wave] = incl;
  __syncthreads();
  if (on) {
    uint32_t run = incl - tot;
    for (int w2 = 0; w2 < wave; ++w2) run += wsum[w2];
#pragma unroll
    for (int u = 0; u < PER; ++u) { cnt[tid * PER + u] = run; run += v[u]; }
  }
  __syncthreads();
}

template <typename T> struct Pair;
template <> struct Pair<float> { using type = float2; };
template <> struct Pair<double> { using type = double2; };
template <typename T> using T2_t = typename Pair<T>::type;

// k = k z + t with the coefficient t taken straight from its scalar register: left to itself the
// compiler builds each step as 3 moves + v_fmac (64 VALU instructions per 16-column row instead of
// 16; measured 113 VALU instructions per point in the 2-D kernel, which made it VALU bound).
__device__ __forceinline__ float fma_sgpr(float k, float z, float t) {
  float r;
  asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(k), "v"(z), "s"(t));
  return r;
}
__device__ __forceinline__ double fma_sgpr(double k, double z, double t) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(k), "v"(z), "s"(t));
  return r;
}

// ---- stacks of tiles (r05; nufft_dense3.hip has the description): descriptor decoding shared by the spread and interp kernels
struct StackDesc {
  int col, z0, nz, p0, p1;   // column, first tile in z, tiles; piece: its points [p0, p1), else p0 < 0
};
__device__ __forceinline__ StackDesc stack_load(const int4* __restrict__ segs, int s) {
  const int4 v = segs[s];
  StackDesc d;
  d.col = v.x; d.z0 = v.y & 0xffff; d.nz = v.y >> 16; d.p0 = v.z; d.p1 = v.w;
  return d;
}
struct StackColumn { int item, t0, t1; };
__device__ __forceinline__ StackColumn stack_column(const Geom& g, int col) {
  const int ncol_item = g.ntile[0] * g.ntile[1];
  StackColumn c;
  c.item = col / ncol_item;
  const int r = col - c.item * ncol_item;
  c.t1 = r / g.ntile[0];
  c.t0 = r - c.t1 * g.ntile[0];
  return c;
}
__device__ __forceinline__ int stack_tile_index(const Geom& g, const StackColumn& c, int t2) {
  const int tc[3] = {c.t0, c.t1, t2};
  return c.item * g.ntiles_item + tile_id(g, tc);
}

}  // namespace
}  // namespace nufft_hip
