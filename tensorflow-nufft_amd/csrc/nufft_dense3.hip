// 3-D type-1 spreading for kernel widths 2..6 in single precision (tol >= 1e-4; BASELINE config 4),
// packed fixed-point accumulation in LDS -- the kernel behind launch_spread for fixed-point plans.
// Replaces the reference's SpreadSubproblem3D* kernels (nufft_plan.cu.cc:1295-1511: one thread per
// point, w^3 pairs of shared-memory float atomics each).
//
// What the r02 kernel (spread_wave3_kernel<FX>) was bound by, measured (profiles/r03_pmc_cfg4.txt):
// 65 VALU instructions per point -- the VALU busy for the whole 11.7 ms -- and 6 ds_add_u64 per point
// with 36 of 64 lanes carrying a stencil cell. This kernel changes both:
//
//  * Lanes cover the stencil densely. A wave-instruction adds a W x YB x ZB block of cells (6 x 3 x 3
//    = 54 lanes at W = 6; 4 x 4 x 4 = 64 at W = 4), and ceil(W / YB) * ceil(W / ZB) of them make a
//    point: 4 ds_add_u64 at W = 6 (6 before), 1 at W <= 4 (W before). Which lane carries which cell
//    is a compile-time table built so that each of the four 16-lane groups an LDS atomic is served in
//    touches 16 distinct 8-byte LDS columns (32 banks) whatever the point's position: row and plane
//    strides are searched for, cells are dealt to the groups by their column, and the idle lanes (10 at
//    W = 6) add 0 at the columns their group leaves free.
//  * float -> packed fixed point costs two instructions per atomic instead of six: v_pk_fma_f32 onto
//    1.5 * 2^23 leaves round-to-nearest(product) in the low mantissa bits of both halves of a register
//    pair; read as one 64-bit integer that is (B + n_re) 2^32 + (B + n_im), B = 0x4B400000, and one
//    v_lshl_add_u64 with the constant -B (2^32 + 1) turns it into n_re 2^32 + n_im: the sign extension
//    of the low field folded into the high one, ready for ds_add_u64.
//  * Kernel values reach the lanes through LDS as 16-byte reads that serve TWO points: per point the
//    lane needs kx[x] (re c, im c), ky[y'], ky[y' + YB], kz[z'], kz[z' + ZB] -- three ds_read_b128 per
//    point pair; no v_readlane broadcasts in the loop except the point's tile offset.
//
// What bounds it (EXPERIMENTS.md section 4b, profiles/r03_pmc_cfg4*.txt): the VGPR -> LDS data path, 2 cycles per source
// dword of an LDS write or atomic: 4 x 7.0 (ds_add_u64) + 4.5 (12 ds_write_b64 per 16 staged points) = 33 cycles per
// point and CU at W = 6 (measured main loop: 30 with two workgroups per CU), against ~12 VALU instructions.
//
// Accumulation format (unchanged from r01/r02, EXPERIMENTS.md section 4): one 64-bit integer per fine cell
// holding (re, im) as two signed 32-bit fields in units of `step`, chosen per subproblem so that no
// cell can overflow (sum of max(|re c|, |im c|) of the subproblem's strengths <= 2^31 steps) and no
// single contribution leaves the exact range of the FMA conversion (|n| < 2^22; a dominant strength is converted
// with v_cvt behind the loop instead of coarsening the step for the others). Tiles with more
// subproblems than Geom::fx_max_subs are left to the fp64-plane kernels (nufft_kernels.hip).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "nufft_device.h"
#include "nufft_hip_internal.h"

namespace nufft_hip {

namespace {

constexpr int kDenseTile = 16;   // tile edge in x and y (the plan's 3-D tiles: 16 x 16 x 8)
#ifndef NUFFT_DENSE_NW   // (experiment builds)
#define NUFFT_DENSE_NW 12
#endif
constexpr int kDenseNW = NUFFT_DENSE_NW;     // waves per workgroup (two workgroups per CU)

// ---- compile-time lane layout -------------------------------------------------------------------

constexpr int dense_yb(int W) { return W == 4 ? 4 : (W == 2 ? 2 : 3); }
constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }

struct DenseLayout {
  int ls, ps;          // row / plane stride of the LDS tile in 8-byte elements
  int rows, planes;    // allocated rows per plane and planes (incl. the rows / planes zero lanes spill into)
  int cell[64];        // element offset of the lane's cell relative to the point's start cell
  int sx[64], sy[64], sz[64];   // staging slots the lane reads: kx (W = the zero slot), ky, kz
  bool ok;
};

// An LDS atomic of 8 bytes per lane is served in four groups of 16 consecutive lanes, each across 32 banks
// (measured, tools/ubench/lds_pattern_bench.hip: 16-lane groups on the same 16 columns do not conflict, two
// 8-lane groups on the same 8 columns do): a group is conflict free when its 16 cells lie in 16 distinct
// columns, column = 8-byte element index mod 16. Strides: no column may hold more than 4 of the block's cells.
constexpr bool dense_strides_ok(int W, int YB, int ZB, int ls, int ps) {
  int cnt[16] = {};
  for (int z = 0; z < ZB; ++z)
    for (int y = 0; y < YB; ++y)
      for (int x = 0; x < W; ++x) {
        const int c = (x + y * ls + z * ps) & 15;
        if (++cnt[c] > 4) return false;
      }
  return true;
}

constexpr DenseLayout make_dense_layout(int W, int TZ) {
  DenseLayout L{};
  const int YB = dense_yb(W), ZB = YB;
  const int NYH = cdiv(W, YB), NZH = cdiv(W, ZB);
  const int L0 = kDenseTile + W - 1;
  L.rows = kDenseTile - 1 + YB * NYH;     // >= tile + W - 1
  L.planes = TZ - 1 + ZB * NZH;
  L.ok = false;
  for (int ls = L0; ls < L0 + 12 && !L.ok; ++ls) {
    for (int pad = 0; pad < 16 && !L.ok; ++pad) {
      const int ps = ls * L.rows + pad;
      if (dense_strides_ok(W, YB, ZB, ls, ps)) {
        L.ls = ls;
        L.ps = ps;
        L.ok = true;
      }
    }
  }
  if (!L.ok) return L;
  // deal the cells to the four 16-lane groups: one cell per column and group
  bool used[4][16] = {};
  int n[4] = {0, 0, 0, 0};
  int lane_of_next[4] = {0, 16, 32, 48};
  for (int z = 0; z < ZB; ++z)
    for (int y = 0; y < YB; ++y)
      for (int x = 0; x < W; ++x) {
        const int cell = x + y * L.ls + z * L.ps;
        const int col = cell & 15;
        int best = -1;
        for (int grp = 0; grp < 4; ++grp)
          if (!used[grp][col] && (best < 0 || n[grp] < n[best])) best = grp;
        if (best < 0) { L.ok = false; return L; }
        used[best][col] = true;
        ++n[best];
        const int lane = lane_of_next[best]++;
        L.cell[lane] = cell;
        L.sx[lane] = x; L.sy[lane] = y; L.sz[lane] = z;
      }
  // idle lanes: add 0 at a column their group leaves free
  for (int grp = 0; grp < 4; ++grp) {
    int col = 0;
    for (int lane = lane_of_next[grp]; lane < 16 * (grp + 1); ++lane) {
      while (used[grp][col]) ++col;
      used[grp][col] = true;
      L.cell[lane] = col;
      L.sx[lane] = W;   // the zero slot
      L.sy[lane] = 0; L.sz[lane] = 0;
    }
  }
  return L;
}

template <int W, int TZ> struct DenseCfg {
  static constexpr DenseLayout L = make_dense_layout(W, TZ);
  static_assert(L.ok, "no conflict-free LDS strides found for this width");
  static constexpr int YB = dense_yb(W), ZB = YB;
  static constexpr int NYH = cdiv(W, YB), NZH = cdiv(W, ZB);
  static constexpr int NW = kDenseNW;
  static constexpr int SLOTS = W + 1 + YB + ZB;  // 16-byte staging slots per point pair
  static constexpr int HALF = 16;               // points staged at a time per wave (64-point chunks stay in registers;
                                                // 16 keep the workgroup at 66 KB of LDS at W = 6: two per CU)
  // (+ room for the idle lanes' columns behind the last cell; even: the staging area behind it is read 16 bytes at
  // a time, and a ds_read_b128 at an address that is not a multiple of 16 costs ~60 stall cycles)
  static constexpr int plane_elems = (L.ps * L.planes + 64 + 1) & ~1;
  static constexpr size_t stage_bytes = (size_t)NW * (HALF / 2) * SLOTS * 16;
  static constexpr size_t lds_bytes = (size_t)plane_elems * 8 + stage_bytes + 2 * NW * sizeof(float) + 64;
  // GROUP: + the permutation of the in-LDS sort by start cell (its counters borrow the plane before it is zeroed)
  static constexpr int NKEY = kDenseTile * kDenseTile * TZ;       // start cells of a tile
  static constexpr int kMaxSub = 4096;                            // points per subproblem (the plan's cap)
  static constexpr size_t lds_bytes_group = lds_bytes + kMaxSub * 2 + 64;
  static_assert(NKEY * 4 <= plane_elems * 8, "sort counters must fit in the plane");
};

// per-lane tables in constant memory (one 16-byte load per lane at kernel start)
struct LaneTab { int cell[64]; int sx[64]; int sy[64]; int sz[64]; };
template <int W, int TZ>
constexpr LaneTab make_lane_tab() {
  LaneTab t{};
  constexpr DenseLayout L = DenseCfg<W, TZ>::L;
  for (int i = 0; i < 64; ++i) { t.cell[i] = L.cell[i]; t.sx[i] = L.sx[i]; t.sy[i] = L.sy[i]; t.sz[i] = L.sz[i]; }
  return t;
}
template <int W, int TZ> __constant__ const LaneTab kLaneTab = make_lane_tab<W, TZ>();

// Piecewise-polynomial kernel values of the first W stencil cells in three dimensions (coefficient
// loop outside: one wave-uniform row load feeds 3 W independent FMAs; cf. horner8, nufft_kernels.hip).
constexpr int kDenseCoef = 10;
template <int W>
__device__ __forceinline__ void horner3(const float* __restrict__ tab, int nc, float z0, float z1, float z2,
                                        float (&k0)[W], float (&k1)[W], float (&k2)[W]) {
  if (nc <= kDenseCoef) {
#pragma unroll
    for (int q = 0; q < W; ++q) { const float t = tab[(kDenseCoef - 1) * kMaxW + q]; k0[q] = t; k1[q] = t; k2[q] = t; }
#pragma unroll
    for (int k = kDenseCoef - 2; k >= 0; --k) {
#pragma unroll
      for (int q = 0; q < W; ++q) {
        const float t = tab[k * kMaxW + q];
        k0[q] = fmaf(k0[q], z0, t); k1[q] = fmaf(k1[q], z1, t); k2[q] = fmaf(k2[q], z2, t);
      }
    }
  } else {
#pragma unroll
    for (int q = 0; q < W; ++q) { const float t = tab[(nc - 1) * kMaxW + q]; k0[q] = t; k1[q] = t; k2[q] = t; }
    for (int k = nc - 2; k >= 0; --k) {
#pragma unroll
      for (int q = 0; q < W; ++q) {
        const float t = tab[k * kMaxW + q];
        k0[q] = fmaf(k0[q], z0, t); k1[q] = fmaf(k1[q], z1, t); k2[q] = fmaf(k2[q], z2, t);
      }
    }
  }
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
#ifndef NUFFT_DENSE_EXP   // (experiment builds, tools/dense_loop_experiment.sh: pieces of dense3_accumulate left out -- wrong results, timing
#define NUFFT_DENSE_EXP 0 //  only; 1 no LDS atomics, 2 no staging reads, 4 no kernel evaluation / staging writes)
#endif

// Phase timestamps (experiment build -DNUFFT_HIP_PHASE_LOG, tools/phase_log_experiment.sh; compiled out otherwise)
#ifdef NUFFT_HIP_PHASE_LOG
constexpr int kPhaseSlots3 = 8, kPhaseLogWgs3 = 65536;
__device__ unsigned long long g_phase_log3[kPhaseLogWgs3 * kPhaseSlots3];
#define NUFFT_PHASE3(k) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < kPhaseLogWgs3) g_phase_log3[blockIdx.x * kPhaseSlots3 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define NUFFT_PHASE3(k) do { } while (0)
#endif

// The accumulation loop of spread_dense3_kernel: this wave's share [wbeg, wend) of the npt points starting at record
// p0 (through perm when GROUP), see the kernel's header comment. stage: the wave's staging area (its zero slot set).
template <int W, int TZ, bool FUSED, bool GROUP>
__device__ __forceinline__ void dense3_accumulate(const SortedPoints<float>& sp, const float2* __restrict__ cc, int p0, int wbeg,
                                                  int wend, float pre, float big_limit, const float* __restrict__ horner, int nc,
                                                  unsigned char* stage, unsigned long long* plane, const uint16_t* perm, int lane) {
  using C = DenseCfg<W, TZ>;
  constexpr int LS = C::L.ls, PS = C::L.ps, YB = C::YB, ZB = C::ZB, NYH = C::NYH, NZH = C::NZH;
  constexpr int SLOTS = C::SLOTS, HALF = C::HALF;
  const FusedRec3* rec3 = reinterpret_cast<const FusedRec3*>(sp.rec);
  const int cell_b = kLaneTab<W, TZ>.cell[lane] * 8;
  const int rd_x = kLaneTab<W, TZ>.sx[lane] * 16;
  const int rd_y = (W + 1 + kLaneTab<W, TZ>.sy[lane]) * 16;
  const int rd_z = (W + 1 + YB + kLaneTab<W, TZ>.sz[lane]) * 16;
  const v2f magic = {12582912.f, 12582912.f};                 // 1.5 * 2^23
  const unsigned long long unbias = 0ull - 0x4B4000004B400000ull;
  unsigned char* plane_b = reinterpret_cast<unsigned char*>(plane);
  // (Loading chunk i + 1's records and strengths before chunk i's atomics was measured, r03: no gain at config 4 --
  // 7.75 ms either way, the other 23 waves of the CU already cover a wave's load latency -- and 10 % slower on a
  // sparse set, 256^3 with M = 1e7. Running the first chunk's phase 1 before the barrier the step waits behind moved
  // 2 k cycles from the main loop into that phase and left the workgroup at 122 k. Neither kept: the LDS data path,
  // shared by the CU's two workgroups, is busy either way.)
  for (int base = wbeg; base < wend; base += 64) {
    // phase 1: one point per lane -- record, strength, 3 W kernel values
    const int js = base + lane;
    int j = p0 + (js < wend ? js : wend - 1);
    if constexpr (GROUP) j = p0 + (int)perm[js < wend ? js : wend - 1];
    int off = 0;
    float kx[W], ky[W], kz[W];
    float cre = 0.f, cim = 0.f, bre = 0.f, bim = 0.f;   // (bre, bim): a strength too large for the FMA conversion
#pragma unroll
    for (int q = 0; q < W; ++q) { kx[q] = 0.f; ky[q] = 0.f; kz[q] = 0.f; }
    if (js < wend) {
      float2 cv;
      PointView<float> rec;
      if constexpr (FUSED) {
        const FusedRec3 f = rec3[j];   // two 16-byte loads
        rec = unpack_rec<float, 3>(f.r);
        cv = make_float2(f.re, f.im);
      } else {
        rec = unpack_rec<float, 3>(sp.rec[j]);
        cv = cc[rec.idx];
      }
      cre = cv.x * pre;
      cim = cv.y * pre;
      if (fmaxf(fabsf(cre), fabsf(cim)) > big_limit) { bre = cre; bim = cim; cre = 0.f; cim = 0.f; }
      off = ((int)(rec.loc & 1023) + (int)((rec.loc >> 10) & 1023) * LS + (int)((rec.loc >> 20) & 1023) * PS) * 8;
      horner3<W>(horner, nc, rec.z0, rec.z1, rec.z2, kx, ky, kz);
    }
    int left = wend - base;
    if (left > 64) left = 64;
    // GROUP: a point ends a run when the next one starts in another cell (or the chunk / the wave's share ends)
    unsigned long long tailm = 0ull;
    if constexpr (GROUP) {
      const int off_next = __shfl_down(off, 1);
      tailm = __ballot(js < wend && (lane == 63 || js == wend - 1 || off_next != off));
    }
    unsigned long long acc[NZH * NYH];
#pragma unroll
    for (int k = 0; k < NZH * NYH; ++k) acc[k] = 0ull;
    int nrun = 0;
#pragma unroll
    for (int h = 0; h < 64 / HALF; ++h) {
      if (h * HALF >= left) break;
      // stage HALF points: pair p holds points 2p, 2p + 1 of this round side by side in every 16-byte slot
      if (!(NUFFT_DENSE_EXP & 4) && lane / HALF == h) {
        unsigned char* s = stage + (((lane % HALF) >> 1) * SLOTS) * 16 + (lane & 1) * 8;
#pragma unroll
        for (int x = 0; x < W; ++x) *reinterpret_cast<v2f*>(s + x * 16) = (v2f){kx[x] * cim, kx[x] * cre};
#pragma unroll
        for (int y = 0; y < YB; ++y)
          *reinterpret_cast<v2f*>(s + (W + 1 + y) * 16) = (v2f){ky[y], (NYH > 1 && y + YB < W) ? ky[y + YB] : 0.f};
#pragma unroll
        for (int z = 0; z < ZB; ++z)
          *reinterpret_cast<v2f*>(s + (W + 1 + YB + z) * 16) = (v2f){kz[z], (NZH > 1 && z + ZB < W) ? kz[z + ZB] : 0.f};
      }
      int npts = left - h * HALF;
      if (npts > HALF) npts = HALF;
      // phase 2: every lane adds its cell of every point; two points per staging read
#pragma unroll 4
      for (int p = 0; p < HALF / 2; ++p) {
        if (2 * p >= npts) break;
#if NUFFT_DENSE_EXP & 2
        const v4f rx = {(float)p, 1.f, 2.f, (float)lane}, ry = {1.f, (float)p, 3.f, 2.f}, rz = {2.f, 1.f, (float)p, 1.f};
#else
        const v4f rx = *reinterpret_cast<const v4f*>(stage + p * SLOTS * 16 + rd_x);
        const v4f ry = *reinterpret_cast<const v4f*>(stage + p * SLOTS * 16 + rd_y);
        const v4f rz = *reinterpret_cast<const v4f*>(stage + p * SLOTS * 16 + rd_z);
#endif
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const v2f kxc = u ? (v2f){rx.z, rx.w} : (v2f){rx.x, rx.y};
          const v2f kyv = u ? (v2f){ry.z, ry.w} : (v2f){ry.x, ry.y};
          const v2f kzv = u ? (v2f){rz.z, rz.w} : (v2f){rz.x, rz.y};
          const int q = h * HALF + 2 * p + u;
          if constexpr (!GROUP) {
            unsigned char* dst = plane_b + __builtin_amdgcn_readlane(off, q) + cell_b;
#pragma unroll
            for (int zh = 0; zh < NZH; ++zh) {
              const float kzq = zh ? kzv.y : kzv.x;
              const v2f yz = kyv * (v2f){kzq, kzq};
#pragma unroll
              for (int yh = 0; yh < NYH; ++yh) {
                const float f = yh ? yz.y : yz.x;
                const v2f fx = __builtin_elementwise_fma(kxc, (v2f){f, f}, magic);
                const unsigned long long x = __builtin_bit_cast(unsigned long long, fx) + unbias;
#if NUFFT_DENSE_EXP & 1
                if (x == 0x123456789ull) *reinterpret_cast<unsigned long long*>(dst) = 1ull;
#else
                atomicAdd(reinterpret_cast<unsigned long long*>(dst + (yh * YB * LS + zh * ZB * PS) * 8), x);
#endif
              }
            }
          } else {
#pragma unroll
            for (int zh = 0; zh < NZH; ++zh) {
              const float kzq = zh ? kzv.y : kzv.x;
              const v2f yz = kyv * (v2f){kzq, kzq};
#pragma unroll
              for (int yh = 0; yh < NYH; ++yh) {
                const float f = yh ? yz.y : yz.x;
                const v2f fx = __builtin_elementwise_fma(kxc, (v2f){f, f}, magic);
                acc[zh * NYH + yh] += __builtin_bit_cast(unsigned long long, fx);   // (mod 2^64; the biases go at the flush)
              }
            }
            ++nrun;
            if ((tailm >> q) & 1ull) {   // wave-uniform
              unsigned char* dst = plane_b + __builtin_amdgcn_readlane(off, q) + cell_b;
              const unsigned long long unb = (unsigned long long)nrun * unbias;
#pragma unroll
              for (int zh = 0; zh < NZH; ++zh)
#pragma unroll
                for (int yh = 0; yh < NYH; ++yh) {
#if NUFFT_DENSE_EXP & 1
                  if (acc[zh * NYH + yh] + unb == 0x123456789ull) *reinterpret_cast<unsigned long long*>(dst) = 1ull;
#else
                  atomicAdd(reinterpret_cast<unsigned long long*>(dst + (yh * YB * LS + zh * ZB * PS) * 8), acc[zh * NYH + yh] + unb);
#endif
                  acc[zh * NYH + yh] = 0ull;
                }
              nrun = 0;
            }
          }
        }
      }
    }
    // strengths above 2^22 steps (only where one strength dominates its subproblem): one at a time, staged alone as
    // the first point of pair 0, converted with v_cvt (exact for the whole 32-bit field)
    unsigned long long pend = __ballot(bre != 0.f || bim != 0.f);
    while (pend) {
      const int src = __ffsll((long long)pend) - 1;
      pend &= pend - 1;
      if (lane == src) {
#pragma unroll
        for (int x = 0; x < W; ++x) *reinterpret_cast<v2f*>(stage + x * 16) = (v2f){kx[x] * bim, kx[x] * bre};
#pragma unroll
        for (int y = 0; y < YB; ++y)
          *reinterpret_cast<v2f*>(stage + (W + 1 + y) * 16) = (v2f){ky[y], (NYH > 1 && y + YB < W) ? ky[y + YB] : 0.f};
#pragma unroll
        for (int z = 0; z < ZB; ++z)
          *reinterpret_cast<v2f*>(stage + (W + 1 + YB + z) * 16) = (v2f){kz[z], (NZH > 1 && z + ZB < W) ? kz[z + ZB] : 0.f};
      }
      const v4f rx = *reinterpret_cast<const v4f*>(stage + rd_x);
      const v4f ry = *reinterpret_cast<const v4f*>(stage + rd_y);
      const v4f rz = *reinterpret_cast<const v4f*>(stage + rd_z);
      const v2f kxc = {rx.x, rx.y}, kyv = {ry.x, ry.y}, kzv = {rz.x, rz.y};
      unsigned char* dst = plane_b + __builtin_amdgcn_readlane(off, src) + cell_b;
#pragma unroll
      for (int zh = 0; zh < NZH; ++zh) {
        const float kzq = zh ? kzv.y : kzv.x;
        const v2f yz = kyv * (v2f){kzq, kzq};
#pragma unroll
        for (int yh = 0; yh < NYH; ++yh) {
          const float f = yh ? yz.y : yz.x;
          const v2f pr = kxc * (v2f){f, f};
          const int ii = __float2int_rn(pr.x), ir = __float2int_rn(pr.y);
          const unsigned long long x = ((unsigned long long)(unsigned)(ir + (ii >> 31)) << 32) | (unsigned)ii;
          atomicAdd(reinterpret_cast<unsigned long long*>(dst + (yh * YB * LS + zh * ZB * PS) * 8), x);
        }
      }
    }
  }
}

// FUSED: the records are FusedRec3 (strength behind the 16-byte record): nothing is gathered.
// GROUP (dense point sets): the subproblem's points are counting-sorted by stencil start cell in LDS first (as
// spread_2d_w8_group_kernel does), and consecutive points that share a start cell add their packed contributions
// in registers (one v_lshl_add_u64 per atomic saved; the bias of the FMA conversion is taken out once per run)
// before ONE set of ds_add_u64. Pays above ~1.2 points per fine cell (dense3_grouped below has the measurements).
template <int W, int TZ, bool FUSED, bool GROUP>
__global__ __launch_bounds__(kDenseNW * 64) void spread_dense3_kernel(
    Geom g, SortedPoints<float> sp, const float* __restrict__ horner, const float* __restrict__ c,
    float* __restrict__ fw, int64_t c_stride, int64_t fw_stride, float scale) {
  using C = DenseCfg<W, TZ>;
  constexpr int LS = C::L.ls, PS = C::L.ps, NW = C::NW;
  constexpr int L0 = kDenseTile + W - 1, L1 = kDenseTile + W - 1, L2 = TZ + W - 1;
  constexpr int SLOTS = C::SLOTS, HALF = C::HALF;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* plane = reinterpret_cast<unsigned long long*>(smem_raw);
  unsigned char* stage_all = smem_raw + (size_t)C::plane_elems * 8;
  float* red = reinterpret_cast<float*>(stage_all + C::stage_bytes);   // [2 NW]

  int tb, p0, p1, slot, nsub;
  NUFFT_PHASE3(0);
  if (!locate_subproblem(g, sp.tile_start, sp.sub_start, blockIdx.x, &tb, &p0, &p1, &slot, &nsub)) return;
  if (nsub > g.fx_max_subs) return;   // crowded tile: the fp64-plane launches behind this one take it
  NUFFT_PHASE3(1);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: keeps the point loops' bounds in SGPRs)
  const float2* cc = reinterpret_cast<const float2*>(c) + (int64_t)slot * c_stride;
  const FusedRec3* rec3 = reinterpret_cast<const FusedRec3*>(sp.rec);
  const int npt = p1 - p0;
  uint16_t* perm = reinterpret_cast<uint16_t*>(red + 2 * NW + 8);   // [kMaxSub] (GROUP)
  if constexpr (GROUP) {
    // counting sort of the subproblem by start cell: keys 4 + 4 + 3 bits (tile 16 x 16 x 8), counters in the
    // (not yet zeroed) plane, 16-bit permutation; records stay where the global sort put them
    constexpr int NT = NW * 64, NKEY = C::NKEY, KBITS = TZ == 8 ? 11 : 10, IT = (C::kMaxSub + NT - 1) / NT;
    uint32_t* cnt = reinterpret_cast<uint32_t*>(plane);
    uint32_t* wsum = reinterpret_cast<uint32_t*>(perm + C::kMaxSub);
    for (int i = tid; i < NKEY; i += NT) cnt[i] = 0u;
    __syncthreads();
    uint32_t kr[IT];
#pragma unroll
    for (int u = 0; u < IT; ++u) {   // (loads unconditional on clamped indices: issued back to back)
      const int i = tid + u * NT;
      const int ic = p0 + (i < npt ? i : npt - 1);
      uint4 w;
      if constexpr (FUSED) w = *reinterpret_cast<const uint4*>(&rec3[ic].r);
      else w = *reinterpret_cast<const uint4*>(&sp.rec[ic]);
      kr[u] = (w.x >> 28) | ((w.y >> 28) << 4) | ((w.z >> 28) << 8);
    }
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const int i = tid + u * NT;
      if (i < npt) kr[u] |= atomicAdd(&cnt[kr[u]], 1u) << KBITS;
    }
    __syncthreads();
    scan_counts<NT, NKEY>(cnt, wsum, tid);
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const int i = tid + u * NT;
      if (i < npt) perm[cnt[kr[u] & (uint32_t)(NKEY - 1)] + (kr[u] >> KBITS)] = (uint16_t)i;
    }
    __syncthreads();
  }
  NUFFT_PHASE3(2);
  for (int i = tid; i < C::plane_elems; i += NW * 64) plane[i] = 0ull;

  // step of the fixed-point grid (see the header comment). The transform's largest and mean strength are known
  // (cstats_kernel, one streaming pass before this launch): unless one strength dominates (largest > 8 x the mean)
  // the subproblem's sum is bounded by its point COUNT x the largest strength and no pass over its own strengths --
  // a second random gather of every c[idx], r03: 16.6 % of a workgroup's life at config 4 -- is needed.
  // (a NaN strength makes the slot's sum NaN -- fmaxf would drop it from the maximum -- and then the step, i.e. every
  // cell this launch writes: non-finite input gives non-finite output, as it does on the floating-point paths)
  const float sum_g = sp.cstats[2 * slot + 1];
  const float top_g = sum_g == sum_g ? sp.cstats[2 * slot] : sum_g;
  const bool own_pass = top_g * (float)c_stride > 8.f * sum_g;   // (workgroup-uniform)
  float part = 0.f, big = 0.f;
  if (own_pass) {
    for (int j = p0 + tid; j < p1; j += NW * 64) {
      float2 cv;
      if constexpr (FUSED) cv = *reinterpret_cast<const float2*>(&rec3[j].re);
      else cv = cc[sp.rec[j].idx];
      const float m = fmaxf(fabsf(cv.x), fabsf(cv.y));
      part += fmaf(0.f, cv.x + cv.y, m);   // (NaN / Inf components make the sum NaN)
      big = fmaxf(big, m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      part += __shfl_down(part, o);
      big = fmaxf(big, __shfl_down(big, o));
    }
    if (lane == 0) { red[wave] = part; red[NW + wave] = big; }
  }
  // this wave's staging area: zero the kx slot the idle lanes read (never written again)
  unsigned char* stage = stage_all + (size_t)wave * (HALF / 2) * SLOTS * 16;
  if (lane < HALF / 2) *reinterpret_cast<v4f*>(stage + (lane * SLOTS + W) * 16) = (v4f){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  NUFFT_PHASE3(3);
  float bound = (float)npt * top_g, top = top_g;
  if (own_pass) {
    bound = 0.f; top = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) { bound += red[k]; top = fmaxf(top, red[NW + k]); }
  }
  const float amp = fabsf(scale) * g.fx_headroom;   // (the fitted polynomials overshoot 1 slightly)
  const float room = 2147483000.f - (float)npt;     // 2^31 minus the rounding of every contribution
  // Two constraints on the step: no cell may overflow 32 bits (sum rule), and the FMA conversion is exact only
  // for |n| < 2^22 per contribution (top rule). Strengths of similar size (largest <= 8 x the mean): the larger
  // of the two -- the top rule binds for subproblems of fewer than ~512 points and is still 2^-22 of the largest
  // strength. One dominant strength: that step would drown the others, so it follows the mean instead, and the
  // few strengths above 2^22 steps are added behind the loop with the exact conversion (v_cvt, any |n| < 2^31).
  const float s_sum = bound * amp / room;
  const bool skewed = top * (float)npt > 8.f * bound;
  float step = fmaxf(s_sum, (skewed ? 2.f * bound / (float)npt : top) * amp * (1.f / 4194000.f));
  if (!(top == top) || !(bound == bound)) step = top + bound;   // NaN strengths (fmaxf would drop them)
  const float pre = step > 0.f ? scale / step : 0.f;
  const float big_limit = 4194000.f / g.fx_headroom;   // |c| in steps above which a point waits for the exact pass

  const int nc = g.ncoef;
  // (indices into the subproblem, in sorted order when GROUP)
  const int share = (npt + NW - 1) / NW;
  const int wbeg = wave * share;
  const int wend = (wbeg + share < npt) ? wbeg + share : npt;
  dense3_accumulate<W, TZ, FUSED, GROUP>(sp, cc, p0, wbeg, wend, pre, big_limit, horner, nc, stage, plane, perm, lane);
  __syncthreads();
  NUFFT_PHASE3(4);

  // write-out: unpack, scale back, add to the periodic fine grid (consecutive lanes carry (re, im) of
  // consecutive cells: contiguous bytes per wave-instruction)
  int t0, t1, t2;
  tile_coords(g, tb, &t0, &t1, &t2);
  const int o0 = t0 * kDenseTile, o1 = t1 * kDenseTile, o2 = t2 * TZ;
  float* out = fw + 2 * (int64_t)slot * fw_stride;
  for (RowWalk r(wave, L1); r.a2 < L2; r.advance(NW, L1)) {
    const int g1 = wrap1(o1 + r.a1, g.nf[1]);
    const int g2 = wrap1(o2 + r.a2, g.nf[2]);
    const int64_t rowbase = (int64_t)g.nf[0] * (g1 + (int64_t)g.nf[1] * g2);
    const int lrow = r.a2 * PS + r.a1 * LS;
    for (int e = lane; e < 2 * L0; e += 64) {
      const int a0 = e >> 1, comp = e & 1;
      const long long t = (long long)plane[lrow + a0];
      const int im_sum = (int)(unsigned)(t & 0xffffffffll);
      const int re_sum = (int)((t - (long long)im_sum) >> 32);
      const float v = (float)(comp ? im_sum : re_sum) * step;
      if (v != 0.f) glb_add(&out[2 * (rowbase + wrap1(o0 + a0, g.nf[0])) + comp], v);
    }
  }
  NUFFT_PHASE3(5);
}

template <int W, int TZ, bool FUSED, bool GROUP>
hipError_t launch_dense3(const Geom& g, const SortedPoints<float>& sp, const float* horner, const float* c, float* fw,
                         dim3 grid, int64_t c_stride, int64_t fw_stride, float scale, hipStream_t stream) {
  using C = DenseCfg<W, TZ>;
  hipError_t e = hipSuccess;
  constexpr size_t lds = GROUP ? C::lds_bytes_group : C::lds_bytes;
  if (lds > 64 * 1024)
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(spread_dense3_kernel<W, TZ, FUSED, GROUP>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  spread_dense3_kernel<W, TZ, FUSED, GROUP><<<grid, C::NW * 64, lds, stream>>>(g, sp, horner, c, fw, c_stride, fw_stride, scale);
  return hipGetLastError();
}

// ================================================================================================
// Widths 7 and 8 (tol 1e-5 / the default 1e-6): packed fixed point with an EXACT 32-bit conversion and a step
// that comes from a count-filter bound, on the 8 x 8 (x, y) lane patch of spread_wave3_kernel with the staging
// scheme of the kernel above. Replaces the 2 w ds_add_f64 per point of the fp64-plane form (r02) by w ds_add_u64.
//
// Why the r01-r03 rules do not reach tol 1e-6. The output error added by the quantisation is ~ 11 (step / top)
// (top = the largest strength; measured r02/r03: 2.4e-6 at step = top / 2^22), so w = 8 needs step <= ~top 2^-25.
// The FMA conversion of the kernels above is exact to 2^22 steps only, and the sum rule (step = sum of the
// subproblem's strengths / 2^31: no cell can overflow whatever the points do) allows ~64 points per subproblem at
// that step. Two changes:
//
//  * Conversion: v_pk_mul_f32 + 2 v_cvt_rpi_i32_f32, exact for the whole 32-bit field. The sign fix-up of the low
//    (imaginary) field would cost two more instructions per atomic; instead every point is spread with a
//    non-negative imaginary strength -- a point with im c < 0 is negated (both components) and SUBTRACTED
//    (ds_sub_u64; the choice is wave-uniform, one point per pass) -- and the kernel values are clamped at 0 (the
//    fit undershoots by ~1e-9 at the stencil's edge), so the low field of every word is >= 0 and
//    word = n_re 2^32 + n_im needs no fix-up: 3 VALU instructions per atomic.
//  * Step: a cell's sum is bounded by top * sum_i K_i(cell) <= top * B, B = max over the tile's cells of the
//    start-cell COUNTS of the subproblem filtered with the per-tap maxima of the kernel polynomials, separably in
//    x, y, z (bound3_kernel). B depends on the points only, so it is computed once per set_points, and it is ~14
//    at 0.22 points per fine cell, 37 at 0.75 (the sum rule's figure there: 1536): step = top B / 2^31 = top 2^-26
//    at 0.75 points per cell. Subproblems whose B would make the step too coarse for the plan's tolerance
//    (Geom::fx_bound_limit; clustered points) and tiles with more than fx_max_subs subproblems are left to the
//    fp64-plane kernels, flagged by a negative entry of sub_bound.
//    `top` is the largest max(|re c|, |im c|) of the whole transform (cstats_kernel, one streaming pass per
//    launch) unless the strengths are too uneven for the subproblem's B (B x largest / mean strength above
//    2.9 x the limit, kPatchCrest below): then the subproblem takes its own pass over its strengths and uses the
//    smallest of three bounds -- its sum, its largest x B, and the start-cell sums of its strengths filtered with
//    the tap maxima (the count filter with the strengths as weights, run in the plane's LDS before it is zeroed).
//    r04 measurements (64^3 modes, tol 1e-6, error added in quadrature to the kernel's own 2.3e-7): largest x B
//    alone 5.1e-7 for lognormal strengths at B = 37 and 6.8e-7 at B = 48; with the weighted bound 0.7e-7 / 0.8e-7.
//
// LDS: one plane of 24-element rows (24 = 8 mod 16: the two rows of a 16-lane group cover 16 distinct 8-byte
// columns, conflict free for every point position), (16 + W - 1) rows, (8 + W - 1) planes = 66 KB at W = 8, and
// 96 bytes of staged kernel values per point, 8 points per wave at a time: 76 KB, two workgroups per CU.
// The separable filter of bound3_kernel (and of the strength-weighted bound in spread_patch3_kernel): cnt[TZ][T][T + 1]
// start-cell weights -> max over the (T + W - 1)^2 (TZ + W - 1) cells of the weights filtered with the tap maxima km.
// Every thread of the NT-thread workgroup calls it with cnt complete (barrier passed) and gets the maximum;
// a: TZ T L floats, b: TZ L L floats, wmax: NT / 64 floats of scratch (L = T + W - 1).
// x and y passes: cnt[TZ][T][T + 1] -> b[TZ][L][L] (a: TZ T L floats of scratch); ends with a barrier
// (NSLOT > 0: layer z goes to slot (slot0 + z) mod NSLOT of a circular buffer of NSLOT layers -- stack_filter_max)
template <int W, int TZ, int NT, int NSLOT = 0>
__device__ __forceinline__ void count_filter_xy(const uint32_t* cnt, float* a, float* b, const float (&km)[W], int tid, int slot0 = 0) {
  // count rows of 17 words: the x pass reads one LINE per lane, and a lane stride of 16 words would put a wave on 4 banks
  constexpr int T = kDenseTile, L = T + W - 1, CP = T + 1;
  // x: line (z, y) of T counts -> L values
  for (int line = tid; line < TZ * T; line += NT) {
    float in[T];
#pragma unroll
    for (int i = 0; i < T; ++i) in[i] = (float)cnt[line * CP + i];
#pragma unroll
    for (int i = 0; i < L; ++i) {
      float v = 0.f;
#pragma unroll
      for (int t = 0; t < W; ++t)
        if (i - t >= 0 && i - t < T) v = fmaf(km[t], in[i - t], v);
      a[line * L + i] = v;
    }
  }
  __syncthreads();
  // y: line (z, i) of T values -> L values
  for (int line = tid; line < TZ * L; line += NT) {
    const int z = line / L, i = line - z * L;
    int zs = z;
    if constexpr (NSLOT > 0) { zs = slot0 + z; zs = zs >= NSLOT ? zs - NSLOT : zs; }
    float in[T];
#pragma unroll
    for (int y = 0; y < T; ++y) in[y] = a[(z * T + y) * L + i];
#pragma unroll
    for (int j = 0; j < L; ++j) {
      float v = 0.f;
#pragma unroll
      for (int t = 0; t < W; ++t)
        if (j - t >= 0 && j - t < T) v = fmaf(km[t], in[j - t], v);
      b[(zs * L + j) * L + i] = v;
    }
  }
  __syncthreads();
}
// the workgroup's maximum of every thread's `best` (wmax: NT / 64 floats); ends with the value in every thread
template <int NT>
__device__ __forceinline__ float workgroup_max(float best, float* wmax, int tid) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_down(best, o));
  if ((tid & 63) == 0) wmax[tid >> 6] = best;
  __syncthreads();
  float m = wmax[0];
#pragma unroll
  for (int k = 1; k < NT / 64; ++k) m = fmaxf(m, wmax[k]);
  return m;
}
template <int W, int TZ, int NT>
__device__ __forceinline__ float count_filter_max(const uint32_t* cnt, float* a, float* b, float* wmax, const float (&km)[W],
                                                  int tid) {
  constexpr int T = kDenseTile, L = T + W - 1, LZ = TZ + W - 1;
  count_filter_xy<W, TZ, NT>(cnt, a, b, km, tid);
  // z: line (j, i) of TZ values -> maximum of the LZ outputs
  float best = 0.f;
  for (int line = tid; line < L * L; line += NT) {
    float in[TZ];
#pragma unroll
    for (int z = 0; z < TZ; ++z) in[z] = b[z * L * L + line];
#pragma unroll
    for (int k = 0; k < LZ; ++k) {
      float v = 0.f;
#pragma unroll
      for (int t = 0; t < W; ++t)
        if (k - t >= 0 && k - t < TZ) v = fmaf(km[t], in[k - t], v);
      best = fmaxf(best, v);
    }
  }
  return workgroup_max<NT>(best, wmax, tid);
}

// With the step from the transform's largest strength, the quantisation adds ~1.2e-9 B x (largest / rms strength) to
// the output (tools/fx_error_vs_crest.py, profiles/r04_fx_error_vs_crest.txt; uniform strengths: largest / rms = 1.41).
// A subproblem whose B x (largest / MEAN strength, which is >= largest / rms) exceeds kPatchCrest x Geom::fx_bound_limit
// (= the same 0.28 of the tolerance: 3.5e-9 limit = 1.2e-9 x 2.9 limit) bounds its cells from its own strengths instead.
constexpr float kPatchCrest = 2.9f;
constexpr int kPatchLS = 24;
#ifndef NUFFT_PATCH_NW
#define NUFFT_PATCH_NW 12
#endif
#ifndef NUFFT_STACK_ROWS
#define NUFFT_STACK_ROWS 4
#endif
constexpr int kPatchNW = NUFFT_PATCH_NW;
#ifdef NUFFT_PATCH_MINW   // (experiment builds: waves per SIMD the register allocation must leave room for)
#define NUFFT_PATCH_BOUNDS __launch_bounds__(kPatchNW * 64, NUFFT_PATCH_MINW)
#else
#define NUFFT_PATCH_BOUNDS __launch_bounds__(kPatchNW * 64)
#endif
template <int W, int TZ, int HALF> struct PatchCfg {
  static constexpr int L0 = kDenseTile + W - 1, L1 = kDenseTile + W - 1, L2 = TZ + W - 1;
  static constexpr int LS = kPatchLS, PS = kPatchLS * L1;
  // idle lanes (W = 7: dx = 7 or dy = 7) add 0 at their natural patch address: up to row 22, column 22 of the last
  // plane, i.e. up to 22 * 24 + 22 - PS + 1 elements behind it
  static constexpr int plane_elems = (PS * L2 + 64 + 1) & ~1;
  static constexpr size_t stage_bytes = 0;   // (nothing is staged since r05; HALF is unused)
  static constexpr size_t lds_bytes = (size_t)plane_elems * 8 + stage_bytes + 2 * kPatchNW * sizeof(float) + 64;
};

__device__ __forceinline__ int cvt_rpi(float x) {   // floor(x + 0.5) in one instruction
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
typedef __attribute__((address_space(3))) unsigned char lds_byte;

// The accumulation loop of the w = 7, 8 kernels: this wave's points [wbeg, wend) of the records starting at `rec`
// (strengths gathered through cc), every lane adding its (dx, dy) column of every point's stencil, plane after plane.
// r05: nothing is staged in LDS. Each lane evaluates the kernel polynomial of ITS OWN tap pair (dx, dy) for the point
// of the pass: the coefficients of a lane's two taps stay in 20 VGPRs for the whole kernel, the point's Horner
// arguments reach the lanes as scalars (v_readlane), and one v_pk_fma_f32 per term evaluates kx[dx] and ky[dy]
// together (9 per point). Only kz (needed by every lane, plane after plane) is evaluated one point per lane and
// broadcast. The r04 loop staged kx c and ky of 8 points at a time through 9 KB of LDS (8 active lanes per store);
// same arithmetic term for term, bit-identical LDS sums, and the same speed (r05, profiles/r05_stack_ab.txt): the
// loop is bound by VALU issue -- ~51 instructions per point at the ~4.4 cycles per wave-instruction and SIMD the
// chip sustains (tools/ubench/valu_rate_bench.hip) = 56 cycles per point and CU -- beside 8 ds_add_u64 at ~6 = 48.
constexpr int kPatchCoef = kDenseCoef;   // (terms held per lane; plans with more keep the staged loop)
template <int W>
__device__ __forceinline__ void horner1(const float* __restrict__ tab, float z, float (&k)[W]) {
#pragma unroll
  for (int q = 0; q < W; ++q) k[q] = tab[(kPatchCoef - 1) * kMaxW + q];
#pragma unroll
  for (int j = kPatchCoef - 2; j >= 0; --j) {
#pragma unroll
    for (int q = 0; q < W; ++q) k[q] = fmaf(k[q], z, tab[j * kMaxW + q]);
  }
}
// have_first: the record and strength of the wave's FIRST 64 points (lane's point wbeg + lane) were loaded by the caller
// (spread_stack3_kernel fetches the next tile's while it writes this one's planes out).
template <int W, int TZ, int HALF>
__device__ __forceinline__ void patch3_accumulate(const Rec<float>* __restrict__ rec, const float2* __restrict__ cc, int wbeg,
                                                         int wend, float pre, const float* __restrict__ horner,
                                                         const v2f (&coef)[kPatchCoef], lds_byte* plane_l, int lane,
                                                         bool have_first = false, Rec<float> first_rec = Rec<float>(),
                                                         float2 first_c = make_float2(0.f, 0.f)) {
  using C = PatchCfg<W, TZ, HALF>;
  constexpr int LS = C::LS, PS = C::PS;
  const int dx = lane & 7, dy = lane >> 3;
  const int cell_b = (dy * LS + dx) * 8;
  for (int base = wbeg; base < wend; base += 64) {
    // phase 1: one point per lane -- record, strength, the W kernel values in z
    const int js = base + lane;
    int off = 0;
    float kz[8], z0 = 0.f, z1 = 0.f, cre = 0.f, cim = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) kz[q] = 0.f;
    if (js < wend) {
      const bool pre_loaded = have_first && base == wbeg;   // (wave-uniform)
      const PointView<float> pv = unpack_rec<float, 3>(pre_loaded ? first_rec : rec[js]);
      const float2 cv = pre_loaded ? first_c : cc[pv.idx];
      cre = cv.x * pre;
      cim = cv.y * pre;
      off = ((int)(pv.loc & 1023) + (int)((pv.loc >> 10) & 1023) * LS + (int)((pv.loc >> 20) & 1023) * PS) * 8;
      z0 = pv.z0;
      z1 = pv.z1;
      float h2[W];
      horner1<W>(horner, pv.z2, h2);
#pragma unroll
      for (int q = 0; q < W; ++q) kz[q] = fmaxf(h2[q], 0.f);
    }
    // a point with a negative imaginary strength is negated here and subtracted below
    const unsigned long long negm = __ballot(cim < 0.f);
    if (cim < 0.f) { cre = -cre; cim = -cim; }
    int left = wend - base;
    if (left > 64) left = 64;
    // phase 2: a pass per point; every lane evaluates its taps and adds its (dx, dy) column, plane after plane
    for (int q = 0; q < left; ++q) {
      const v2f zz = {bcast_lane(z0, q), bcast_lane(z1, q)};
      v2f k = coef[kPatchCoef - 1];
#pragma unroll
      for (int j = kPatchCoef - 2; j >= 0; --j) k = __builtin_elementwise_fma(k, zz, coef[j]);
      const float kxv = fmaxf(k.x, 0.f), kyv = fmaxf(k.y, 0.f);
      const v2f kxc = (v2f){kxv, kxv} * (v2f){bcast_lane(cim, q), bcast_lane(cre, q)};
      const v2f xy = kxc * (v2f){kyv, kyv};
      lds_u64* dst = (lds_u64*)(plane_l + __builtin_amdgcn_readlane(off, q) + cell_b);
      if ((negm >> q) & 1ull) {   // wave-uniform
#pragma unroll
        for (int dz = 0; dz < W; ++dz) {
          const float kzq = bcast_lane(kz[dz], q);
          const v2f v = xy * (v2f){kzq, kzq};
          const unsigned long long word = ((unsigned long long)(unsigned)cvt_rpi(v.y) << 32) | (unsigned)cvt_rpi(v.x);
          __hip_atomic_fetch_sub(dst + dz * PS, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      } else {
#pragma unroll
        for (int dz = 0; dz < W; ++dz) {
          const float kzq = bcast_lane(kz[dz], q);
          const v2f v = xy * (v2f){kzq, kzq};
          const unsigned long long word = ((unsigned long long)(unsigned)cvt_rpi(v.y) << 32) | (unsigned)cvt_rpi(v.x);
          __hip_atomic_fetch_add(dst + dz * PS, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
      }
    }
  }
}
// the lane's coefficient pairs: term j of tap dx = lane & 7 (x) and of tap dy = lane >> 3 (y); taps >= W: zero
template <int W>
__device__ __forceinline__ void patch3_lane_coef(const float* __restrict__ horner, int lane, v2f (&coef)[kPatchCoef]) {
  const int dx = lane & 7, dy = lane >> 3;
#pragma unroll
  for (int j = 0; j < kPatchCoef; ++j)
    coef[j] = (v2f){dx < W ? horner[j * kMaxW + dx] : 0.f, dy < W ? horner[j * kMaxW + dy] : 0.f};
}

template <int W, int TZ, int HALF>
__global__ NUFFT_PATCH_BOUNDS void spread_patch3_kernel(
    Geom g, SortedPoints<float> sp, const float* __restrict__ horner, const float* __restrict__ c,
    float* __restrict__ fw, int64_t c_stride, int64_t fw_stride, float scale) {
  using C = PatchCfg<W, TZ, HALF>;
  constexpr int LS = C::LS, PS = C::PS, NW = kPatchNW, L0 = C::L0, L1 = C::L1, L2 = C::L2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* plane = reinterpret_cast<unsigned long long*>(smem_raw);
  unsigned char* stage_all = smem_raw + (size_t)C::plane_elems * 8;
  float* red = reinterpret_cast<float*>(stage_all + C::stage_bytes);   // [2 NW]

  int tb, p0, p1, slot, nsub;
  NUFFT_PHASE3(0);
  if (!locate_subproblem(g, sp.tile_start, sp.sub_start, blockIdx.x, &tb, &p0, &p1, &slot, &nsub)) return;
  const float bound_b = sp.sub_bound[blockIdx.x];
  if (bound_b < 0.f) return;   // too crowded for the fixed-point grid: the fp64-plane launches behind this one take it
  NUFFT_PHASE3(1);
  NUFFT_PHASE3(2);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float2* cc = reinterpret_cast<const float2*>(c) + (int64_t)slot * c_stride;
  const int npt = p1 - p0;

  // step of the fixed-point grid (see the header comment)
  const float sum_g = sp.cstats[2 * slot + 1];
  const float top_g = sum_g == sum_g ? sp.cstats[2 * slot] : sum_g;   // (NaN strengths: see spread_dense3_kernel)
  const bool skewed = bound_b * top_g * (float)c_stride > kPatchCrest * g.fx_bound_limit * sum_g;
  float top = top_g, sum = 3.0e38f, cap = top_g * bound_b;
  if (skewed) {
    // (workgroup-uniform) strengths too far from uniform for this subproblem's B: its own strengths decide. Its cells
    // are bounded by the start-cell sums of max(|re c|, |im c|) filtered with the tap maxima -- the count filter of
    // bound3_kernel with the strengths as weights (in units of the transform's largest / 2^19, rounded up) --, by the
    // sum of its strengths, and by its largest one x the count bound. The plane is not in use yet: its LDS holds
    // the filter.
    constexpr int T = kDenseTile, FL = T + W - 1, CP = T + 1;
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem_raw);                 // [TZ][T][CP]
    float* fa = reinterpret_cast<float*>(cnt + TZ * T * CP);               // [TZ][T][FL]
    float* fb = fa + TZ * T * FL;                                          // [TZ][FL][FL]
    float* fmx = fb + TZ * FL * FL;                                        // [NW]
    static_assert((size_t)(TZ * T * CP + TZ * T * FL + TZ * FL * FL + NW) * 4 <= (size_t)C::plane_elems * 8, "filter scratch");
    for (int i = tid; i < TZ * T * CP; i += NW * 64) cnt[i] = 0u;
    __syncthreads();
    // (weights in units of the largest / wscale, rounded up; a start cell holds at most the subproblem's points:
    // wscale (npt + 1) stays below 2^32 whatever max_subproblem_size the plan was given -- r04 advisor)
    float wscale = 524288.f;
    if ((float)(npt + 1) * wscale > 4.0e9f) wscale = floorf(4.0e9f / (float)(npt + 1));
    const float inv = top_g > 0.f ? wscale / top_g : 0.f;
    float part = 0.f, big = 0.f;
    for (int j = p0 + tid; j < p1; j += NW * 64) {
      const PointView<float> rec = unpack_rec<float, 3>(sp.rec[j]);
      const float2 cv = cc[rec.idx];
      const float m = fmaxf(fabsf(cv.x), fabsf(cv.y));
      part += fmaf(0.f, cv.x + cv.y, m);   // (NaN / Inf components make the sum NaN)
      big = fmaxf(big, m);
      const uint32_t wgt = (uint32_t)ceilf(m * inv);   // <= 2^19 + 1; NaN -> 0 (the NaN step below takes over)
      atomicAdd(&cnt[(((rec.loc >> 20) & 1023u) * T + ((rec.loc >> 10) & 1023u)) * CP + (rec.loc & 1023u)], wgt);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      part += __shfl_down(part, o);
      big = fmaxf(big, __shfl_down(big, o));
    }
    if (lane == 0) { red[wave] = part; red[NW + wave] = big; }
    __syncthreads();
    sum = 0.f; top = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) { sum += red[k]; top = fmaxf(top, red[NW + k]); }
    float km[W];
#pragma unroll
    for (int t = 0; t < W; ++t) km[t] = g.fx_tap[t];
    const float wb = count_filter_max<W, TZ, NW * 64>(cnt, fa, fb, fmx, km, tid);
    cap = fminf(top * bound_b, top_g * wb * (1.0001f / wscale));
    __syncthreads();   // (the scratch becomes the plane)
  }
  for (int i = tid; i < C::plane_elems; i += NW * 64) plane[i] = 0ull;
  const float amp = fabsf(scale) * g.fx_headroom;
  const float room = 2147483000.f - (float)npt;     // 2^31 minus one step of rounding per contribution
  // (no finer than top 2^-29: every contribution stays below 2^29 steps)
  float step = fmaxf(fminf(sum, cap), top * 1.8626451e-9f * room) * amp / room;
  if (!(top == top) || !(sum == sum)) step = top + sum;   // NaN strengths (fminf / fmaxf would drop them)
  const float pre = step > 0.f ? scale / step : 0.f;
  __syncthreads();
  NUFFT_PHASE3(3);

  const int share = (npt + NW - 1) / NW;
  const int wbeg = wave * share;
  const int wend = (wbeg + share < npt) ? wbeg + share : npt;
  v2f coef[kPatchCoef];
  patch3_lane_coef<W>(horner, lane, coef);
  patch3_accumulate<W, TZ, HALF>(sp.rec + p0, cc, wbeg, wend, pre, horner, coef, (lds_byte*)plane, lane);
  __syncthreads();
  NUFFT_PHASE3(4);

  // write-out: unpack, scale back, add to the periodic fine grid
  int t0, t1, t2;
  tile_coords(g, tb, &t0, &t1, &t2);
  const int o0 = t0 * kDenseTile, o1 = t1 * kDenseTile, o2 = t2 * TZ;
  float* out = fw + 2 * (int64_t)slot * fw_stride;
  for (RowWalk r(wave, L1); r.a2 < L2; r.advance(NW, L1)) {
    const int g1 = wrap1(o1 + r.a1, g.nf[1]);
    const int g2 = wrap1(o2 + r.a2, g.nf[2]);
    const int64_t rowbase = (int64_t)g.nf[0] * (g1 + (int64_t)g.nf[1] * g2);
    const int lrow = r.a2 * PS + r.a1 * LS;
    for (int e = lane; e < 2 * L0; e += 64) {
      const int a0 = e >> 1, comp = e & 1;
      const long long t = (long long)plane[lrow + a0];
      const int im_sum = (int)(unsigned)(t & 0xffffffffll);
      const int re_sum = (int)((t - (long long)im_sum) >> 32);
      const float v = (float)(comp ? im_sum : re_sum) * step;
      if (v != 0.f) glb_add(&out[2 * (rowbase + wrap1(o0 + a0, g.nf[0])) + comp], v);
    }
  }
  NUFFT_PHASE3(5);
}

// ---- count-filter bound of every subproblem (set_points of a patch plan) -----------------------------------------
// B = max over the cells of the tile (+ halo) of sum_t0 sum_t1 sum_t2 N(i - t0, j - t1, k - t2) kmax[t0] kmax[t1] kmax[t2],
// N = the subproblem's points per start cell: every cell sum of the spread is at most (largest strength) x B.
// One workgroup per subproblem; the three filter passes keep a line's inputs in registers.
#ifndef NUFFT_BOUND_THREADS
#define NUFFT_BOUND_THREADS 512   // (tools/ab_bound_threads.sh, 65536 subproblems: 256 threads 361 us, 384 466, 512 331)
#endif
constexpr int kBoundThreads = NUFFT_BOUND_THREADS;
// (Measured r04, 65536 subproblems of 458 points: this form 0.36-0.38 ms before the count rows were padded. A form
// with two lines per thread in v_pk_fma_f32 -- 40 % fewer wave-instructions -- ran 0.49-0.69 ms: a phase then keeps
// one or two waves of the workgroup busy and the kernel is bound by the latency of each phase, not by VALU issue.)
template <int W, int TZ>
__global__ __launch_bounds__(kBoundThreads) void bound3_kernel(Geom g, const Rec<float>* __restrict__ rec, int rec_stride,
                                                              const int32_t* __restrict__ tile_start,
                                                              const int32_t* __restrict__ sub_start, TapMax taps,
                                                              float* __restrict__ sub_bound, int* __restrict__ fb_list) {
  constexpr int T = kDenseTile, L = T + W - 1, CP = T + 1, NT = kBoundThreads;
  __shared__ uint32_t cnt[TZ * T * CP];
  __shared__ float a[TZ * T * L];     // [z][y][i]  (L is odd: conflict-free line-per-lane writes)
  __shared__ float b[TZ * L * L];     // [z][j][i]
  __shared__ float wmax[NT / 64];
  int tb, p0, p1, slot, nsub;
  if (!locate_subproblem(g, tile_start, sub_start, blockIdx.x, &tb, &p0, &p1, &slot, &nsub)) return;
  const int tid = threadIdx.x, npt = p1 - p0;
  if (nsub > g.fx_max_subs) {   // (every further subproblem of a tile adds its share of quantisation noise)
    if (tid == 0) {
      sub_bound[blockIdx.x] = -1.f;
      fb_list[1 + atomicAdd(&fb_list[0], 1)] = (int)blockIdx.x;
    }
    return;
  }
  if (npt <= 16) {   // (B <= the point count; nothing finer is needed of so few points)
    if (tid == 0) sub_bound[blockIdx.x] = (float)(npt > 0 ? npt : 1);
    return;
  }
  for (int i = tid; i < TZ * T * CP; i += NT) cnt[i] = 0u;
  __syncthreads();
  for (int j = p0 + tid; j < p1; j += NT) {
    const Rec<float>& r = rec_at(rec, j, rec_stride);
    const uint32_t l0 = r.loc >> 28, l1 = __float_as_uint(r.z0) >> 28, l2 = __float_as_uint(r.z1) >> 28;
    atomicAdd(&cnt[(l2 * T + l1) * CP + l0], 1u);
  }
  __syncthreads();
  float km[W];
#pragma unroll
  for (int t = 0; t < W; ++t) km[t] = taps.k[t];
  float m = count_filter_max<W, TZ, NT>(cnt, a, b, wmax, km, tid);
  if (tid == 0) {
    m *= 1.0001f;   // (float sums of non-negative terms)
    if (m > g.fx_bound_limit) {
      sub_bound[blockIdx.x] = -m;
      fb_list[1 + atomicAdd(&fb_list[0], 1)] = (int)blockIdx.x;
    } else {
      sub_bound[blockIdx.x] = m;
    }
  }
}

// ================================================================================================
// Stacks (r05): the w = 7, 8 kernel over runs of tiles that are consecutive in z.
//
// A subproblem of spread_patch3_kernel writes its tile + halo, (16 + W - 1)^2 (8 + W - 1) cells = 3.9 x the tile at
// W = 8, to the fine grid with float atomics; the memory side takes those at ~1.3 TB/s (one 64-byte request per
// ~50 ps chip-wide, MI355X_MICROARCH.md: global float atomics), and below ~0.3 points per fine cell that write-out,
// not the LDS accumulation, is the kernel (r04: 456 us of 0.65 ms for the reference harness's 3-D case).
// The halo in z is the largest part of it (15 planes for 8). A STACK is a run of tiles of one (x, y) tile column,
// z0 .. z0 + nz - 1, spread by one workgroup: after tile t its 8 finished planes are written out, the W - 1 halo
// planes MOVE DOWN to become the first planes of tile t + 1 and the freed planes are zeroed -- one pass over the
// plane in LDS, what the zero-fill of a fresh subproblem costs anyway. Cells written per tile: 23 x 23 x (8 + 7 / nz)
// instead of 23 x 23 x 15 at W = 8: 2.1-2.3 x the fine grid instead of 3.9 x.
//
// The packed fixed-point planes carry over from tile to tile, so a stack has ONE step: from its count-filter bound
// (the filter of bound3_kernel run over the stack's tiles with the last W - 1 start-cell layers of the previous tile
// carried into the z pass: the exact maximum over every cell the stack writes) and, for uneven strengths, from the
// same filter over the strengths. Stacks are cut so that none holds more than `cap` points (workgroups stay of
// similar length whatever the density) or more than `len` tiles; a tile with more than max_sub points is cut into
// the subproblems locate_subproblem would give it, each a stack of its own (PIECES: one tile, a point range).
// Flagged stacks (bound above Geom::fx_bound_limit, tiles of more than fx_max_subs pieces) go to the fp64-plane
// launches behind this one as the subproblems they consist of.
// One wave per tile column: cuts the column into stacks (greedy along z), two passes -- count, take a range of
// descriptor slots with one atomic, write. The order of the descriptors depends on arrival; nothing else does.
constexpr int kStackPlanWaves = 4;
__global__ __launch_bounds__(kStackPlanWaves * 64) void stack_plan_kernel(Geom g, const int32_t* __restrict__ tile_start,
                                                                         int ncol, int cap, int len, int4* __restrict__ segs,
                                                                         int* __restrict__ seg_count, int max_segs) {
  const int lane = threadIdx.x & 63;
  const int col = blockIdx.x * kStackPlanWaves + (threadIdx.x >> 6);
  if (col >= ncol) return;
  const StackColumn cc = stack_column(g, col);
  const int ntz = g.ntile[2], S = g.max_sub;
  int base = 0;
  for (int pass = 0; pass < 2; ++pass) {
    int nseg = 0, first = 0, last = 0, pts = 0;   // open stack: tiles [first, last], `pts` points (0: none open)
    for (int zb = 0; zb < ntz; zb += 64) {
      int n = 0, b = 0;
      if (zb + lane < ntz) {
        const int t = stack_tile_index(g, cc, zb + lane);
        b = tile_start[t];
        n = tile_start[t + 1] - b;
      }
      const int m = ntz - zb < 64 ? ntz - zb : 64;
      for (int i = 0; i < m; ++i) {
        const int ni = __shfl(n, i), bi = __shfl(b, i), tz = zb + i;   // (wave-uniform)
        if (ni == 0) continue;   // (empty tiles neither open nor end a stack; the length rule below counts them)
        const bool big = ni > S;
        if (pts > 0 && (big || pts + ni > cap || tz - first + 1 > len)) {
          if (pass && lane == 0 && base + nseg < max_segs) segs[base + nseg] = make_int4(col, first | ((last - first + 1) << 16), -1, -1);
          ++nseg;
          pts = 0;
        }
        if (big) {
          const int k = (ni + S - 1) / S, sz = (ni + k - 1) / k;
          if (pass)
            for (int j = lane; j < k; j += 64) {
              const int a = bi + j * sz, e = a + sz < bi + ni ? a + sz : bi + ni;
              if (base + nseg + j < max_segs) segs[base + nseg + j] = make_int4(col, tz | (1 << 16), a, e);
            }
          nseg += k;
        } else {
          if (pts == 0) first = tz;
          last = tz;
          pts += ni;
        }
      }
    }
    if (pts > 0) {
      if (pass && lane == 0 && base + nseg < max_segs) segs[base + nseg] = make_int4(col, first | ((last - first + 1) << 16), -1, -1);
      ++nseg;
    }
    if (pass == 0) {
      if (nseg == 0) return;
      if (lane == 0) base = atomicAdd(seg_count, nseg);
      base = __shfl(base, 0);
    }
  }
}

// Filter maximum over a stack: the start-cell weights of every tile through the x and y passes of the count filter,
// the z pass over the tile's TZ layers and the last W - 1 layers of the tile before it. add_tile(i, cnt) adds the
// weights of the stack's i-th tile to cnt[TZ][T][T + 1] (zeroed, behind a barrier; called by every thread).
// a: TZ T L floats, b: (TZ + W - 1) L L floats, wmax: NT / 64 floats. Returns the maximum in every thread.
struct NoPrefetch { __device__ __forceinline__ void operator()(int) const {} };
// pre_tile(i): called before the filter passes of tile i - 1 (and once up front with 0) -- a place to request tile i's
// point range and first records so that their latency hides behind those passes (r06)
template <int W, int TZ, int NT, typename AddTile, typename PreTile = NoPrefetch>
__device__ __forceinline__ float stack_filter_max(int nz, AddTile add_tile, uint32_t* cnt, float* a, float* b, float* wmax,
                                                  const float (&km)[W], int tid, PreTile pre_tile = PreTile()) {
  constexpr int T = kDenseTile, L = T + W - 1, CP = T + 1, H = W - 1, NS = TZ + H;
  static_assert(H <= TZ, "the z halo must end inside the next tile");
  // b: a circular buffer of TZ + H layers -- tile i's layer z in slot (i TZ + z) mod NS, so that the last H layers of
  // tile i - 1 sit in the H slots in front of tile i's (52 KB with cnt and a: three workgroups per CU; two full
  // buffers of TZ layers made 54 KB, two per CU)
  float best = 0.f;
  int s0 = 0;   // slot of this tile's layer 0
  pre_tile(0);
  for (int i = 0; i <= nz; ++i) {   // (i == nz: the halo planes behind the last tile)
    if (i < nz) {
      for (int q = tid; q < TZ * T * CP; q += NT) cnt[q] = 0u;
      __syncthreads();
      add_tile(i, cnt);
      if (i + 1 < nz) pre_tile(i + 1);
      __syncthreads();
      count_filter_xy<W, TZ, NT, NS>(cnt, a, b, km, tid, s0);
    }
    for (int line = tid; line < L * L; line += NT) {
      float in[H + TZ];   // layers TZ - H .. TZ - 1 of the previous tile, then this tile's
#pragma unroll
      for (int m = 0; m < H; ++m) {
        int sl = s0 - H + m;
        sl = sl < 0 ? sl + NS : sl;
        in[m] = i > 0 ? b[sl * L * L + line] : 0.f;
      }
#pragma unroll
      for (int z = 0; z < TZ; ++z) {
        int sl = s0 + z;
        sl = sl >= NS ? sl - NS : sl;
        in[H + z] = i < nz ? b[sl * L * L + line] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < TZ; ++k) {
        if (i == nz && k >= H) break;
        float v = 0.f;
#pragma unroll
        for (int t = 0; t < W; ++t) v = fmaf(km[t], in[H + k - t], v);
        best = fmaxf(best, v);
      }
    }
    // (the next tile's y pass overwrites only slots this z pass is done with: the TZ slots behind its own H)
    s0 += TZ;
    s0 = s0 >= NS ? s0 - NS : s0;
    if (i < nz) __syncthreads();
  }
  return workgroup_max<NT>(best, wmax, tid);
}

template <int W, int TZ>
__global__ __launch_bounds__(kBoundThreads) void bound3_stack_kernel(Geom g, const Rec<float>* __restrict__ rec, int rec_stride,
                                                                    const int32_t* __restrict__ tile_start,
                                                                    const int32_t* __restrict__ sub_start, TapMax taps,
                                                                    const int4* __restrict__ segs, const int* __restrict__ seg_count,
                                                                    float* __restrict__ seg_bound, int* __restrict__ fb_list) {
  constexpr int T = kDenseTile, L = T + W - 1, CP = T + 1, NT = kBoundThreads;
  __shared__ uint32_t cnt[TZ * T * CP];
  __shared__ float a[TZ * T * L];
  __shared__ float b[(TZ + W - 1) * L * L];   // (circular: stack_filter_max)
  __shared__ float wmax[NT / 64];
  const int s = blockIdx.x, tid = threadIdx.x;
  if (s >= seg_count[0]) return;
  const StackDesc d = stack_load(segs, s);
  const StackColumn cc = stack_column(g, d.col);
  if (d.p0 >= 0) {
    // a piece of a tile with more than max_sub points = subproblem `chunk` of that tile
    const int t = stack_tile_index(g, cc, d.z0);
    const int s0 = sub_start[t], k = sub_start[t + 1] - s0;
    if (k > g.fx_max_subs) {   // (every further subproblem of a tile adds its share of quantisation noise)
      if (tid == 0) {
        const int n = tile_start[t + 1] - tile_start[t], sz = (n + k - 1) / k;
        seg_bound[s] = -1.f;
        fb_list[1 + atomicAdd(&fb_list[0], 1)] = s0 + (d.p0 - tile_start[t]) / sz;
      }
      return;
    }
  }
  float km[W];
#pragma unroll
  for (int t = 0; t < W; ++t) km[t] = taps.k[t];
  // r06: the tiles' point ranges are read up front and every thread's first record of tile i + 1 is requested before
  // the filter passes of tile i: the two dependent loads at the head of a tile (range, then records) were exposed
  // once per tile -- at ~76 points per tile most of this kernel's time (sort_cell stage 0.40 ms at 256^3 / M = 1e7)
  constexpr int kRng = 32;
  __shared__ int rng[2 * kRng];
  if (d.p0 < 0) {
    for (int i = tid; i < d.nz && i < kRng; i += NT) {
      const int t = stack_tile_index(g, cc, d.z0 + i);
      rng[2 * i] = tile_start[t];
      rng[2 * i + 1] = tile_start[t + 1];
    }
  }
  __syncthreads();
  auto range_of = [&](int i, int* q0, int* q1) {
    if (d.p0 >= 0) { *q0 = d.p0; *q1 = d.p1; return; }
    if (i < kRng) { *q0 = rng[2 * i]; *q1 = rng[2 * i + 1]; return; }
    const int t = stack_tile_index(g, cc, d.z0 + i);
    *q0 = tile_start[t];
    *q1 = tile_start[t + 1];
  };
  uint32_t nx0 = 0u, nx1 = 0u, nx2 = 0u;   // the start-cell bits of this thread's first record of the next tile
  float m = stack_filter_max<W, TZ, NT>(d.nz, [&](int i, uint32_t* c) {
    int p0, p1;
    range_of(i, &p0, &p1);
    if (p0 + tid < p1) atomicAdd(&c[((nx2 >> 28) * T + (nx1 >> 28)) * CP + (nx0 >> 28)], 1u);   // (requested by pre_tile(i))
    for (int j = p0 + tid + NT; j < p1; j += NT) {
      const Rec<float>& r = rec_at(rec, j, rec_stride);
      const uint32_t l0 = r.loc >> 28, l1 = __float_as_uint(r.z0) >> 28, l2 = __float_as_uint(r.z1) >> 28;
      atomicAdd(&c[(l2 * T + l1) * CP + l0], 1u);
    }
  }, cnt, a, b, wmax, km, tid, [&](int i) {
    int p0, p1;
    range_of(i, &p0, &p1);
    if (p0 + tid < p1) {
      const Rec<float>& r = rec_at(rec, p0 + tid, rec_stride);
      nx0 = r.loc; nx1 = __float_as_uint(r.z0); nx2 = __float_as_uint(r.z1);
    }
  });
  if (tid == 0) {
    m *= 1.0001f;   // (float sums of non-negative terms)
    if (m > g.fx_bound_limit) {
      seg_bound[s] = -m;
      if (d.p0 >= 0) {
        const int t = stack_tile_index(g, cc, d.z0);
        const int s0 = sub_start[t], k = sub_start[t + 1] - s0;
        const int n = tile_start[t + 1] - tile_start[t], sz = (n + k - 1) / k;
        fb_list[1 + atomicAdd(&fb_list[0], 1)] = s0 + (d.p0 - tile_start[t]) / sz;
      } else {
        for (int i = 0; i < d.nz; ++i) {   // (tiles of at most max_sub points: one subproblem each, none when empty)
          const int t = stack_tile_index(g, cc, d.z0 + i);
          if (sub_start[t + 1] > sub_start[t]) fb_list[1 + atomicAdd(&fb_list[0], 1)] = sub_start[t];
        }
      }
    } else {
      seg_bound[s] = m > 1.f ? m : 1.f;
    }
  }
}

// The w = 7, 8 kernel over a stack: finished planes are added to the fine grid with float atomics. (A plain-store form --
// slabs gathered by a merge kernel or by the first FFT pass -- was built and removed in r05: EXPERIMENTS.md section 11.3.)
template <int W, int TZ, int HALF>
__global__ NUFFT_PATCH_BOUNDS void spread_stack3_kernel(
    Geom g, SortedPoints<float> sp, const float* __restrict__ horner, const float* __restrict__ c,
    float* __restrict__ fw, int64_t c_stride, int64_t fw_stride, float scale) {
  using C = PatchCfg<W, TZ, HALF>;
  constexpr int LS = C::LS, PS = C::PS, NW = kPatchNW, L0 = C::L0, L1 = C::L1, L2 = C::L2, H = W - 1, NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* plane = reinterpret_cast<unsigned long long*>(smem_raw);
  unsigned char* stage_all = smem_raw + (size_t)C::plane_elems * 8;
  float* red = reinterpret_cast<float*>(stage_all + C::stage_bytes);   // [2 NW]

  const int s = blockIdx.x;
#ifdef NUFFT_HIP_PHASE_LOG
  const unsigned long long ph_start = __builtin_readcyclecounter();
#endif
  if (s >= sp.seg_count[0]) return;
  const float bound_b = sp.seg_bound[s];
  if (bound_b < 0.f) return;   // flagged: its subproblems are on the fallback list of the fp64-plane launches
  const StackDesc d = stack_load(sp.segs, s);
  const StackColumn col = stack_column(g, d.col);
  const int slot = col.item * (int)gridDim.y + (int)blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float2* cc = reinterpret_cast<const float2*>(c) + (int64_t)slot * c_stride;
  // points of the whole stack (tiles of a column need not be neighbours in the sorted order)
  int npt_all = 0;
  if (d.p0 >= 0) npt_all = d.p1 - d.p0;
  else
    for (int i = 0; i < d.nz; ++i) {
      const int t = stack_tile_index(g, col, d.z0 + i);
      npt_all += sp.tile_start[t + 1] - sp.tile_start[t];
    }

  // step of the fixed-point grid (spread_patch3_kernel; here one for the whole stack)
  const float sum_g = sp.cstats[2 * slot + 1];
  const float top_g = sum_g == sum_g ? sp.cstats[2 * slot] : sum_g;   // (NaN strengths: see spread_dense3_kernel)
  const bool skewed = bound_b * top_g * (float)c_stride > kPatchCrest * g.fx_bound_limit * sum_g;
  float top = top_g, sum = 3.0e38f, cap = top_g * bound_b;
  if (skewed) {
    // (workgroup-uniform) the stack's own strengths decide: start-cell sums of max(|re c|, |im c|) through the stack
    // filter (weights in units of the transform's largest / wscale, rounded up; a start cell holds at most max_sub
    // points: wscale (max_sub + 1) < 2^32), the sum of its strengths, its largest one x the count bound
    constexpr int T = kDenseTile, FL = T + W - 1, CP = T + 1;
    uint32_t* cnt = reinterpret_cast<uint32_t*>(smem_raw);                 // [TZ][T][CP]
    float* fa = reinterpret_cast<float*>(cnt + TZ * T * CP);               // [TZ][T][FL]
    float* fb = fa + TZ * T * FL;                                          // [TZ + W - 1][FL][FL] (circular)
    float* fmx = fb + (TZ + W - 1) * FL * FL;                              // [NW]
    static_assert((size_t)(TZ * T * CP + TZ * T * FL + (TZ + W - 1) * FL * FL + NW) * 4 <= (size_t)C::plane_elems * 8, "filter scratch");
    float wscale = 524288.f;
    if ((float)(g.max_sub + 1) * wscale > 4.0e9f) wscale = floorf(4.0e9f / (float)(g.max_sub + 1));
    const float inv = top_g > 0.f ? wscale / top_g : 0.f;
    float part = 0.f, big = 0.f;
    float km[W];
#pragma unroll
    for (int t = 0; t < W; ++t) km[t] = g.fx_tap[t];
    const float wb = stack_filter_max<W, TZ, NT>(d.nz, [&](int i, uint32_t* cn) {
      int p0 = d.p0, p1 = d.p1;
      if (p0 < 0) {
        const int t = stack_tile_index(g, col, d.z0 + i);
        p0 = sp.tile_start[t];
        p1 = sp.tile_start[t + 1];
      }
      for (int j = p0 + tid; j < p1; j += NT) {
        const PointView<float> rec = unpack_rec<float, 3>(sp.rec[j]);
        const float2 cv = cc[rec.idx];
        const float m = fmaxf(fabsf(cv.x), fabsf(cv.y));
        part += fmaf(0.f, cv.x + cv.y, m);   // (NaN / Inf components make the sum NaN)
        big = fmaxf(big, m);
        const uint32_t wgt = (uint32_t)ceilf(m * inv);   // NaN -> 0 (the NaN step below takes over)
        atomicAdd(&cn[(((rec.loc >> 20) & 1023u) * T + ((rec.loc >> 10) & 1023u)) * CP + (rec.loc & 1023u)], wgt);
      }
    }, cnt, fa, fb, fmx, km, tid);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      part += __shfl_down(part, o);
      big = fmaxf(big, __shfl_down(big, o));
    }
    if (lane == 0) { red[wave] = part; red[NW + wave] = big; }
    __syncthreads();
    sum = 0.f; top = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) { sum += red[k]; top = fmaxf(top, red[NW + k]); }
    cap = fminf(top * bound_b, top_g * wb * (1.0001f / wscale));
    __syncthreads();   // (the scratch becomes the plane)
  }
  for (int i = tid; i < C::plane_elems; i += NT) plane[i] = 0ull;
  const float amp = fabsf(scale) * g.fx_headroom;
  const float room = 2147483000.f - (float)npt_all;     // 2^31 minus one step of rounding per contribution
  float step = fmaxf(fminf(sum, cap), top * 1.8626451e-9f * room) * amp / room;
  if (!(top == top) || !(sum == sum)) step = top + sum;   // NaN strengths (fminf / fmaxf would drop them)
  const float pre = step > 0.f ? scale / step : 0.f;
  __syncthreads();

#ifdef NUFFT_HIP_PHASE_LOG
  // (experiment build) thread 0's cycles by phase, summed over the stack's tiles: [0] setup, then per tile
  // accumulate (its own wave) / wait for the others / write-out + move / barrier behind it
  unsigned long long ph_acc[5] = {0, 0, 0, 0, 0}, ph_prev = __builtin_readcyclecounter();
  ph_acc[0] = ph_prev - ph_start;
#define NUFFT_STACK_PH(k) do { const unsigned long long now_ = __builtin_readcyclecounter(); if ((k) > 0) ph_acc[k] += now_ - ph_prev; else ph_acc[0] += 0; ph_prev = now_; } while (0)
#else
#define NUFFT_STACK_PH(k) do { } while (0)
#endif
  v2f coef[kPatchCoef];
  patch3_lane_coef<W>(horner, lane, coef);
  const int o0 = col.t0 * kDenseTile, o1 = col.t1 * kDenseTile;
  float* out = fw + 2 * (int64_t)slot * fw_stride;
  // The record and the strength of every lane's first point of a tile are fetched a tile ahead: the record while the
  // tile before is accumulated, the strength (through the record's index) before that tile's planes are written out --
  // at a few dozen points per tile the two dependent loads at the head of a tile were ~20 % of its time.
  auto tile_range = [&](int i, int* p0, int* p1) {
    *p0 = d.p0; *p1 = d.p1;
    if (d.p0 < 0) {
      const int t = stack_tile_index(g, col, d.z0 + i);
      *p0 = sp.tile_start[t];
      *p1 = sp.tile_start[t + 1];
    }
  };
  auto first_index = [&](int p0, int p1) {   // the lane's first point of a tile, or -1
    const int npt = p1 - p0, share = (npt + NW - 1) / NW;
    const int wbeg = wave * share, wend = (wbeg + share < npt) ? wbeg + share : npt;
    return wbeg + lane < wend ? p0 + wbeg + lane : -1;
  };
  int q0, q1;
  tile_range(0, &q0, &q1);
  int jn = first_index(q0, q1);
  Rec<float> rec_n = sp.rec[jn >= 0 ? jn : 0];   // (lanes without a point: some valid record)
  float2 c_n = cc[jn >= 0 ? rec_n.idx : 0];
  for (int i = 0; i < d.nz; ++i) {
    const int p0 = q0, p1 = q1;
    const Rec<float> rec_i = rec_n;
    const float2 c_i = c_n;
    const bool more = i + 1 < d.nz;
    if (more) {
      tile_range(i + 1, &q0, &q1);
      jn = first_index(q0, q1);
      rec_n = sp.rec[jn >= 0 ? jn : 0];
    }
    const int npt = p1 - p0;
    const int share = (npt + NW - 1) / NW;
    const int wbeg = wave * share;
    const int wend = (wbeg + share < npt) ? wbeg + share : npt;
    NUFFT_STACK_PH(0);
    patch3_accumulate<W, TZ, HALF>(sp.rec + p0, cc, wbeg, wend, pre, horner, coef, (lds_byte*)plane, lane, true, rec_i, c_i);
    NUFFT_STACK_PH(1);
    if (more) c_n = cc[jn >= 0 ? rec_n.idx : 0];
    __syncthreads();
    NUFFT_STACK_PH(2);
    // tile d.z0 + i is complete in its first TZ planes (the last tile of the stack: in all of them): unpack, scale
    // back, add to the periodic fine grid (consecutive lanes carry (re, im) of consecutive cells); then the halo
    // planes move down to be the first planes of the next tile and what they leave is zeroed
    const bool last = i == d.nz - 1;
    const int o2 = (d.z0 + i) * TZ;
    const int nplanes = last ? L2 : TZ;
    // (R rows per wave at a time: the LDS reads of a batch are issued together -- with the sibling workgroup's
    // atomics queued in front of every LDS access, a row at a time cost ~700 cycles per row, r05 phase log)
    constexpr int R = NUFFT_STACK_ROWS;
    const int nrows = nplanes * L1;
    const int e = lane < 2 * L0 ? lane : 2 * L0 - 1, a0 = e >> 1, comp = e & 1;
    const bool active = lane < 2 * L0;
    const int gx = wrap1(o0 + a0, g.nf[0]);
    for (int r0 = wave; r0 < nrows; r0 += NW * R) {
      int lrow[R], a2v[R];
      int64_t gbase[R];
      long long t[R], hi[R];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int rho = r0 + u * NW < nrows ? r0 + u * NW : r0;   // (a batch's rows past the end repeat its first)
        const int a2 = rho / L1, a1 = rho - a2 * L1;
        a2v[u] = a2;
        lrow[u] = a2 * PS + a1 * LS;
        gbase[u] = (int64_t)g.nf[0] * (wrap1(o1 + a1, g.nf[1]) + (int64_t)g.nf[1] * wrap1(o2 + a2, g.nf[2]));
        t[u] = (long long)plane[lrow[u] + a0];
      }
      if (!last) {
#pragma unroll
        for (int u = 0; u < R; ++u) hi[u] = a2v[u] < H ? (long long)plane[lrow[u] + TZ * PS + a0] : 0ll;
      }
#pragma unroll
      for (int u = 0; u < R; ++u) {
        if (u > 0 && r0 + u * NW >= nrows) break;
        const int im_sum = (int)(unsigned)(t[u] & 0xffffffffll);
        const int re_sum = (int)((t[u] - (long long)im_sum) >> 32);
        const float v = (float)(comp ? im_sum : re_sum) * step;
        if (active && v != 0.f) glb_add(&out[2 * (gbase[u] + gx) + comp], v);
      }
      if (!last && active && comp == 0) {   // (both lanes of a cell have read it: LDS operations of a wave complete in order)
#pragma unroll
        for (int u = 0; u < R; ++u) {
          if (u > 0 && r0 + u * NW >= nrows) break;
          plane[lrow[u] + a0] = (unsigned long long)hi[u];
          if (a2v[u] < H) plane[lrow[u] + TZ * PS + a0] = 0ull;
        }
      }
    }
    NUFFT_STACK_PH(3);
    if (!last) __syncthreads();
    NUFFT_STACK_PH(4);
  }
#ifdef NUFFT_HIP_PHASE_LOG
  if (tid == 0 && blockIdx.y == 0 && blockIdx.x < kPhaseLogWgs3) {
    unsigned long long* lg = g_phase_log3 + (size_t)blockIdx.x * kPhaseSlots3;
    lg[0] = ph_start; lg[1] = ph_acc[0]; lg[2] = ph_acc[1]; lg[3] = ph_acc[2]; lg[4] = ph_acc[3]; lg[5] = ph_acc[4];
    lg[6] = __builtin_readcyclecounter(); lg[7] = (unsigned long long)d.nz;
  }
#endif
}

// spread_dense3_kernel (w <= 6) over a stack (r05): the same z carry as spread_stack3_kernel, without a bound kernel --
// the step follows the r03 rules on the points that can share a cell, i.e. of two neighbouring tiles of the stack
// (sum rule: the largest such pair x the launch's largest strength) and the 2^22 range of the FMA conversion (top rule;
// it binds at most of the densities stacks are taken at: below 0.12 points per cell a pair of tiles holds < 512 points). One dominant strength
// (largest > 8 x the mean): the stack's own sum and largest, as a subproblem's. Pieces of tiles with more than
// fx_max_subs of them are the fp64-plane launches' (crowded_list_kernel lists them as the subproblems they are).
template <int W, int TZ, bool FUSED>
__global__ __launch_bounds__(kDenseNW * 64) void spread_dense3_stack_kernel(
    Geom g, SortedPoints<float> sp, const float* __restrict__ horner, const float* __restrict__ c,
    float* __restrict__ fw, int64_t c_stride, int64_t fw_stride, float scale) {
  using C = DenseCfg<W, TZ>;
  constexpr int LS = C::L.ls, PS = C::L.ps, NW = C::NW, NT = NW * 64, H = W - 1;
  constexpr int L0 = kDenseTile + W - 1, L1 = kDenseTile + W - 1, L2 = TZ + W - 1;
  constexpr int SLOTS = C::SLOTS, HALF = C::HALF;
  static_assert(H <= TZ, "the z halo must end inside the next tile");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned long long* plane = reinterpret_cast<unsigned long long*>(smem_raw);
  unsigned char* stage_all = smem_raw + (size_t)C::plane_elems * 8;
  float* red = reinterpret_cast<float*>(stage_all + C::stage_bytes);   // [2 NW]

  const int s = blockIdx.x;
  if (s >= sp.seg_count[0]) return;
  const StackDesc d = stack_load(sp.segs, s);
  const StackColumn col = stack_column(g, d.col);
  const int slot = col.item * (int)gridDim.y + (int)blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float2* cc = reinterpret_cast<const float2*>(c) + (int64_t)slot * c_stride;
  const FusedRec3* rec3 = reinterpret_cast<const FusedRec3*>(sp.rec);
  // points of the stack, and the most that two neighbouring tiles hold together (what can share a cell)
  int npt_all = 0, pair = 0;
  if (d.p0 >= 0) {
    const int t = stack_tile_index(g, col, d.z0);
    if (sp.sub_start[t + 1] - sp.sub_start[t] > g.fx_max_subs) return;   // crowded tile: the fp64-plane launches take it
    npt_all = pair = d.p1 - d.p0;
  } else {
    int prev = 0;
    for (int i = 0; i < d.nz; ++i) {
      const int t = stack_tile_index(g, col, d.z0 + i);
      const int n = sp.tile_start[t + 1] - sp.tile_start[t];
      npt_all += n;
      pair = pair > prev + n ? pair : prev + n;
      prev = n;
    }
  }
  for (int i = tid; i < C::plane_elems; i += NT) plane[i] = 0ull;

  // step of the fixed-point grid (spread_dense3_kernel's rules)
  const float sum_g = sp.cstats[2 * slot + 1];
  const float top_g = sum_g == sum_g ? sp.cstats[2 * slot] : sum_g;
  const bool own_pass = top_g * (float)c_stride > 8.f * sum_g;   // (workgroup-uniform)
  float part = 0.f, big = 0.f;
  if (own_pass) {
    for (int i = 0; i < d.nz; ++i) {
      int p0 = d.p0, p1 = d.p1;
      if (p0 < 0) {
        const int t = stack_tile_index(g, col, d.z0 + i);
        p0 = sp.tile_start[t];
        p1 = sp.tile_start[t + 1];
      }
      for (int j = p0 + tid; j < p1; j += NT) {
        float2 cv;
        if constexpr (FUSED) cv = *reinterpret_cast<const float2*>(&rec3[j].re);
        else cv = cc[sp.rec[j].idx];
        const float m = fmaxf(fabsf(cv.x), fabsf(cv.y));
        part += fmaf(0.f, cv.x + cv.y, m);   // (NaN / Inf components make the sum NaN)
        big = fmaxf(big, m);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      part += __shfl_down(part, o);
      big = fmaxf(big, __shfl_down(big, o));
    }
    if (lane == 0) { red[wave] = part; red[NW + wave] = big; }
  }
  unsigned char* stage = stage_all + (size_t)wave * (HALF / 2) * SLOTS * 16;
  if (lane < HALF / 2) *reinterpret_cast<v4f*>(stage + (lane * SLOTS + W) * 16) = (v4f){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  float bound = (float)pair * top_g, top = top_g;
  float cnt = (float)pair;
  if (own_pass) {
    bound = 0.f; top = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) { bound += red[k]; top = fmaxf(top, red[NW + k]); }
    cnt = (float)npt_all;
  }
  const float amp = fabsf(scale) * g.fx_headroom;
  const float room = 2147483000.f - (float)npt_all;
  const float s_sum = bound * amp / room;
  const bool skewed = top * cnt > 8.f * bound;
  float step = fmaxf(s_sum, (skewed ? 2.f * bound / cnt : top) * amp * (1.f / 4194000.f));
  if (!(top == top) || !(bound == bound)) step = top + bound;   // NaN strengths (fmaxf would drop them)
  const float pre = step > 0.f ? scale / step : 0.f;
  const float big_limit = 4194000.f / g.fx_headroom;

  const int nc = g.ncoef;
  const int o0 = col.t0 * kDenseTile, o1 = col.t1 * kDenseTile;
  float* out = fw + 2 * (int64_t)slot * fw_stride;
  for (int i = 0; i < d.nz; ++i) {
    int p0 = d.p0, p1 = d.p1;
    if (p0 < 0) {
      const int t = stack_tile_index(g, col, d.z0 + i);
      p0 = sp.tile_start[t];
      p1 = sp.tile_start[t + 1];
    }
    const int npt = p1 - p0;
    const int share = (npt + NW - 1) / NW;
    const int wbeg = wave * share;
    const int wend = (wbeg + share < npt) ? wbeg + share : npt;
    dense3_accumulate<W, TZ, FUSED, false>(sp, cc, p0, wbeg, wend, pre, big_limit, horner, nc, stage, plane, nullptr, lane);
    __syncthreads();
    // write-out of the tile's finished planes and the move of the halo planes: as spread_stack3_kernel
    const bool last = i == d.nz - 1;
    const int o2 = (d.z0 + i) * TZ;
    const int nplanes = last ? L2 : TZ;
    constexpr int R = NUFFT_STACK_ROWS;
    const int nrows = nplanes * L1;
    const int e = lane < 2 * L0 ? lane : 2 * L0 - 1, a0 = e >> 1, comp = e & 1;
    const bool active = lane < 2 * L0;
    const int gx = wrap1(o0 + a0, g.nf[0]);
    for (int r0 = wave; r0 < nrows; r0 += NW * R) {
      int lrow[R], a2v[R];
      int64_t gbase[R];
      long long t[R], hi[R];
#pragma unroll
      for (int u = 0; u < R; ++u) {
        const int rho = r0 + u * NW < nrows ? r0 + u * NW : r0;
        const int a2 = rho / L1, a1 = rho - a2 * L1;
        a2v[u] = a2;
        lrow[u] = a2 * PS + a1 * LS;
        gbase[u] = (int64_t)g.nf[0] * (wrap1(o1 + a1, g.nf[1]) + (int64_t)g.nf[1] * wrap1(o2 + a2, g.nf[2]));
        t[u] = (long long)plane[lrow[u] + a0];
      }
      if (!last) {
#pragma unroll
        for (int u = 0; u < R; ++u) hi[u] = a2v[u] < H ? (long long)plane[lrow[u] + TZ * PS + a0] : 0ll;
      }
#pragma unroll
      for (int u = 0; u < R; ++u) {
        if (u > 0 && r0 + u * NW >= nrows) break;
        const int im_sum = (int)(unsigned)(t[u] & 0xffffffffll);
        const int re_sum = (int)((t[u] - (long long)im_sum) >> 32);
        const float v = (float)(comp ? im_sum : re_sum) * step;
        if (active && v != 0.f) glb_add(&out[2 * (gbase[u] + gx) + comp], v);
      }
      if (!last && active && comp == 0) {
#pragma unroll
        for (int u = 0; u < R; ++u) {
          if (u > 0 && r0 + u * NW >= nrows) break;
          plane[lrow[u] + a0] = (unsigned long long)hi[u];
          if (a2v[u] < H) plane[lrow[u] + TZ * PS + a0] = 0ull;
        }
      }
    }
    if (!last) __syncthreads();
  }
}

// ---- the fp64-plane fallback of the fixed-point plans on 16 x 16 x 8 tiles, cell-grouped (r05) ------------------------
// What set_points leaves to the fp64 planes are subproblems whose count-filter bound is above what the tolerance allows
// (w = 7, 8) and the subproblems of tiles with more than fx_max_subs of them (w <= 6):
// dense clusters -- a kooshball's centre, hundreds of points per start cell. spread_wave3_kernel (one launch per
// component) adds every point's 2 x W plane values with LDS atomics: 0.3 ns per point and launch, 1.75 ms for the 2.9e6
// points of a 256^3 kooshball at M = 3e7 (profiles/r05_kooshball_kernels.txt), a fifth of the transform. Here the
// subproblem's points are counting-sorted by start cell in LDS (4096 at a time; planes persist across the segments),
// every lane evaluates its own (dx, dy) taps as patch3_accumulate does, and a run of points that share a start cell is
// summed in float registers -- W packed FMAs per point -- before ONE set of 2 W ds_add_f64: both fp64 planes in one
// launch (133 KB, one workgroup of 16 waves per CU; the list is short). Runs of one point cost what the old kernel did.
constexpr int kFbNW = 16, kFbSeg = 4096, kFbKeys = kDenseTile * kDenseTile * 8;
constexpr int kFbJoinFrom = 64, kFbJoinMax = 8;   // (as spread_wave3_body: subproblems of a very crowded tile joined per workgroup)
template <int W> struct FbCfg {
  static constexpr int LS = kPatchLS, L0 = kDenseTile + W - 1, L1 = kDenseTile + W - 1, L2 = 8 + W - 1, PS = LS * L1;
  // per component; lanes outside the W x W patch (zero taps) add 0 at their natural 8 x 8 patch address: up to row 22, column
  // 22 of the last plane, i.e. up to 22 * 24 + 22 - PS + 1 elements behind it
  static constexpr int pad = (22 * LS + 22 - PS + 1) > 64 ? (22 * LS + 22 - PS + 1) : 64;
  static constexpr int PE = (PS * L2 + pad + 1) & ~1;
  static constexpr size_t lds_bytes = (size_t)2 * PE * 8 + kFbKeys * 4 + kFbSeg * 2 + 16 * 4 + 64;
};
template <int W>
__device__ __forceinline__ void spread_group3_f64_body(const Geom& g, const SortedPoints<float>& sp, const float* __restrict__ horner,
                                                       const float* __restrict__ c, float* __restrict__ fw, int64_t c_stride,
                                                       int64_t fw_stride, float scale, int sub, int y, int batch) {
  using C = FbCfg<W>;
  constexpr int LS = C::LS, PS = C::PS, L0 = C::L0, L1 = C::L1, L2 = C::L2, PE = C::PE;
  constexpr int NW = kFbNW, NT = NW * 64, IT = kFbSeg / NT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  double* plane_re = reinterpret_cast<double*>(smem_raw);
  double* plane_im = plane_re + PE;
  uint32_t* cnt = reinterpret_cast<uint32_t*>(plane_im + PE);   // [kFbKeys]
  uint16_t* perm = reinterpret_cast<uint16_t*>(cnt + kFbKeys);  // [kFbSeg]
  uint32_t* wsum = reinterpret_cast<uint32_t*>(perm + kFbSeg);  // [16]
  int tb, p0, p1, slot, nsub, chunk, tile_end;
  if (!locate_subproblem(g, sp.tile_start, sp.sub_start, sub, &tb, &p0, &p1, &slot, &nsub, &chunk, &tile_end)) return;
  slot = slot * batch + y;   // (the launch is one workgroup row: `slot` came back as the item; transform y of `batch`)
  if (nsub > kFbJoinFrom) {   // (float atomics of hundreds of write-outs into the same cells: see spread_wave3_body)
    int join = (nsub + kFbJoinFrom - 1) / kFbJoinFrom;
    if (join > kFbJoinMax) join = kFbJoinMax;
    if (chunk % join) return;
    const long long end = (long long)p0 + (long long)join * (p1 - p0);
    p1 = end < (long long)tile_end ? (int)end : tile_end;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 2 * PE; i += NT) plane_re[i] = 0.0;
  const float2* cc = reinterpret_cast<const float2*>(c) + (int64_t)slot * c_stride;
  const int rstride = g.fused ? (int)sizeof(FusedRec3) : (int)sizeof(Rec<float>);
  v2f coef[kPatchCoef];
  patch3_lane_coef<W>(horner, lane, coef);
  const int dx = lane & 7, dy = lane >> 3;
  const bool in_patch = dx < W && dy < W;
  const int cell_b = (dy * LS + dx) * 8;
  for (int seg0 = p0; seg0 < p1; seg0 += kFbSeg) {
    const int n = p1 - seg0 < kFbSeg ? p1 - seg0 : kFbSeg;
    // counting sort of the segment by start cell (4 + 4 + 3 bits of the packed record)
    for (int i = tid; i < kFbKeys; i += NT) cnt[i] = 0u;
    __syncthreads();
    uint32_t kr[IT];
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const int i = tid + u * NT;
      const uint4 w = *reinterpret_cast<const uint4*>(&rec_at(sp.rec, seg0 + (i < n ? i : n - 1), rstride));
      kr[u] = (w.x >> 28) | ((w.y >> 28) << 4) | ((w.z >> 28) << 8);
    }
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const int i = tid + u * NT;
      if (i < n) kr[u] |= atomicAdd(&cnt[kr[u]], 1u) << 11;
    }
    __syncthreads();
    scan_counts<NT, kFbKeys>(cnt, wsum, tid);
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const int i = tid + u * NT;
      if (i < n) perm[cnt[kr[u] & (uint32_t)(kFbKeys - 1)] + (kr[u] >> 11)] = (uint16_t)i;
    }
    __syncthreads();
    const int share = (n + NW - 1) / NW;
    const int wbeg = wave * share;
    const int wend = (wbeg + share < n) ? wbeg + share : n;
    for (int base = wbeg; base < wend; base += 64) {
      // phase 1: one point per lane
      const int js = base + lane;
      int off = -8;
      float kz[8], z0 = 0.f, z1 = 0.f, cre = 0.f, cim = 0.f;
#pragma unroll
      for (int q = 0; q < 8; ++q) kz[q] = 0.f;
      if (js < wend) {
        const PointView<float> pv = unpack_rec<float, 3>(rec_at(sp.rec, seg0 + (int)perm[js], rstride));
        const float2 cv = cc[pv.idx];
        cre = cv.x * scale;
        cim = cv.y * scale;
        off = ((int)(pv.loc & 1023) + (int)((pv.loc >> 10) & 1023) * LS + (int)((pv.loc >> 20) & 1023) * PS) * 8;
        z0 = pv.z0;
        z1 = pv.z1;
        float h2[W];
        horner1<W>(horner, pv.z2, h2);
#pragma unroll
        for (int q = 0; q < W; ++q) kz[q] = h2[q];
      }
      // a point ends a run when the next one starts in another cell (or the chunk / the wave's share ends)
      const int off_next = __shfl_down(off, 1);
      const unsigned long long tailm = __ballot(js < wend && (lane == 63 || js == wend - 1 || off_next != off));
      int left = wend - base;
      if (left > 64) left = 64;
      v2f acc[W];
#pragma unroll
      for (int dz = 0; dz < W; ++dz) acc[dz] = (v2f){0.f, 0.f};
      // phase 2: a pass per point -- the lane's taps, then its (dx, dy) column of the W planes into the run's sums
      for (int q = 0; q < left; ++q) {
        const v2f zz = {bcast_lane(z0, q), bcast_lane(z1, q)};
        v2f k = coef[kPatchCoef - 1];
#pragma unroll
        for (int j = kPatchCoef - 2; j >= 0; --j) k = __builtin_elementwise_fma(k, zz, coef[j]);
        const float kxy = k.x * k.y;
        // (lanes outside the W x W patch have zero taps: their product is forced to 0 rather than computed, so that a NaN /
        // Inf strength stays inside its stencil as on the per-point kernels -- r05 advisor)
        const v2f wc = in_patch ? (v2f){kxy, kxy} * (v2f){bcast_lane(cre, q), bcast_lane(cim, q)} : (v2f){0.f, 0.f};
#pragma unroll
        for (int dz = 0; dz < W; ++dz) {
          const float kzq = bcast_lane(kz[dz], q);
          acc[dz] = __builtin_elementwise_fma(wc, (v2f){kzq, kzq}, acc[dz]);
        }
        if ((tailm >> q) & 1ull) {   // wave-uniform
          const int o = __builtin_amdgcn_readlane(off, q) + cell_b;
          double* pr = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(plane_re) + o);
#pragma unroll
          for (int dz = 0; dz < W; ++dz) {
            lds_add(pr + dz * PS, (double)acc[dz].x);
            lds_add(pr + dz * PS + PE, (double)acc[dz].y);
            acc[dz] = (v2f){0.f, 0.f};
          }
        }
      }
    }
    __syncthreads();   // (the next segment's sort reuses the counters and the permutation)
  }
  // write-out: both planes into the periodic fine grid
  int t0, t1, t2;
  tile_coords(g, tb, &t0, &t1, &t2);
  const int o0 = t0 * kDenseTile, o1 = t1 * kDenseTile, o2 = t2 * 8;
  float* out = fw + 2 * (int64_t)slot * fw_stride;
  for (RowWalk r(wave, L1); r.a2 < L2; r.advance(NW, L1)) {
    const int g1 = wrap1(o1 + r.a1, g.nf[1]);
    const int g2 = wrap1(o2 + r.a2, g.nf[2]);
    const int64_t rowbase = (int64_t)g.nf[0] * (g1 + (int64_t)g.nf[1] * g2);
    const int lrow = r.a2 * PS + r.a1 * LS;
    if (lane < 2 * L0) {
      const int a0 = lane >> 1, comp = lane & 1;
      const float v = (float)plane_re[lrow + a0 + comp * PE];
      if (v != 0.f) glb_add(&out[2 * (rowbase + wrap1(o0 + a0, g.nf[0])) + comp], v);
    }
  }
}
// a small persistent grid walks the list of subproblems that set_points left to the fp64 planes (entries follow the count)
template <int W>
__global__ __launch_bounds__(kFbNW * 64) void spread_group3_f64_kernel(
    Geom g, SortedPoints<float> sp, const float* __restrict__ horner, const float* __restrict__ c,
    float* __restrict__ fw, int64_t c_stride, int64_t fw_stride, float scale, int batch) {
  // (entries cost anything from nothing -- joined subproblems exit -- to a dozen segments: the workgroups draw (entry,
  // transform) pairs from a counter; dealt in strides, a grid of one workgroup per CU ran 6-10 % longer, r05)
  __shared__ int next_it;
  const int n = sp.fb_list[0] * batch;
  for (;;) {
    if (threadIdx.x == 0) next_it = atomicAdd(sp.fb_ticket, 1);
    __syncthreads();
    const int it = next_it;
    __syncthreads();   // (everyone has read it before thread 0 draws again; also: the next subproblem zeroes the planes
                       //  this one's write-out reads)
    if (it >= n) break;
    const int e = it / batch;
    spread_group3_f64_body<W>(g, sp, horner, c, fw, c_stride, fw_stride, scale, sp.fb_list[1 + e], it - e * batch, batch);
  }
}
// ---- strengths of one spread launch: largest and summed max(|re c|, |im c|) per slot ----------------------------
// Two stages, no atomics: a first form had every workgroup add its partial sums to the slot's two floats --
// 4096 atomics on two addresses, serialised at ~60 ns each: 0.24 ms for a 60 us read of 3e7 strengths.
constexpr int kStatsMaxBlocks = 1024;
__global__ __launch_bounds__(256) void cstats_partial_kernel(const float2* __restrict__ c, int64_t M, int64_t c_stride,
                                                              float* __restrict__ partial) {
  const float2* cc = c + (int64_t)blockIdx.y * c_stride;
  float big = 0.f, part = 0.f;
  // (eight loads in flight per thread)
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 7 * stride < M; i += 8 * stride) {
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = cc[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float m = fmaxf(fabsf(v[u].x), fabsf(v[u].y));
      big = fmaxf(big, m);
      part += fmaf(0.f, v[u].x + v[u].y, m);   // (m, or NaN when a component is NaN / Inf: fmaxf alone would drop a NaN)
    }
  }
  for (; i < M; i += stride) {
    const float2 v = cc[i];
    const float m = fmaxf(fabsf(v.x), fabsf(v.y));
    big = fmaxf(big, m);
    part += fmaf(0.f, v.x + v.y, m);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    part += __shfl_down(part, o);
    big = fmaxf(big, __shfl_down(big, o));
  }
  __shared__ float r[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { r[wave] = part; r[4 + wave] = big; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* dst = partial + 2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x);
    dst[0] = fmaxf(fmaxf(r[4], r[5]), fmaxf(r[6], r[7]));
    dst[1] = (r[0] + r[1]) + (r[2] + r[3]);
  }
}
__global__ __launch_bounds__(256) void cstats_final_kernel(const float* __restrict__ partial, int nblk, float* __restrict__ cstats) {
  const float* src = partial + 2 * (size_t)blockIdx.x * nblk;
  float big = 0.f, part = 0.f;
  for (int i = threadIdx.x; i < nblk; i += 256) {
    big = fmaxf(big, src[2 * i]);
    part += src[2 * i + 1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    part += __shfl_down(part, o);
    big = fmaxf(big, __shfl_down(big, o));
  }
  __shared__ float r[8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { r[wave] = part; r[4 + wave] = big; }
  __syncthreads();
  if (threadIdx.x == 0) {
    cstats[2 * blockIdx.x] = fmaxf(fmaxf(r[4], r[5]), fmaxf(r[6], r[7]));
    cstats[2 * blockIdx.x + 1] = (r[0] + r[1]) + (r[2] + r[3]);
  }
}

// ---- shader clock under an LDS-atomic load (bench.py prices the LDS roofline with it) -----------------------------
// Every workgroup runs `iters` conflict-free ds_add_u64 per thread and reports the shader-clock and the
// constant-rate wall-clock ticks that took.
__global__ __launch_bounds__(768) void clock_probe_kernel(unsigned long long* __restrict__ out, int iters) {
  __shared__ unsigned long long cell[768];
  cell[threadIdx.x] = 0ull;
  __syncthreads();
  const unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) atomicAdd(&cell[threadIdx.x], (unsigned long long)i);
  __syncthreads();
  const unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = w1 - w0;
  }
  if (cell[threadIdx.x] == 0xdeadbeefdeadbeefull) out[0] = 0;   // (keeps the loop)
}

}  // namespace

// Shader clock in MHz while every CU runs LDS atomics (~1 ms of them), from the ratio of the two counters;
// scratch: device buffer of >= 2 * 512 unsigned long long. Synchronises the stream.
hipError_t measure_shader_clock_mhz(hipStream_t stream, unsigned long long* scratch, double* mhz) {
  constexpr int kBlocks = 512;
  int dev = 0, wall_khz = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  e = hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, dev);
  if (e != hipSuccess) return e;
  unsigned long long host[2 * kBlocks];
  for (int rep = 0; rep < 2; ++rep) {   // (the first launch warms the clocks up)
    clock_probe_kernel<<<kBlocks, 768, 0, stream>>>(scratch, 20000);
    e = hipMemcpyAsync(host, scratch, sizeof(host), hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
  }
  std::vector<double> r;
  for (int b = 0; b < kBlocks; ++b)
    if (host[2 * b + 1] > 0) r.push_back((double)host[2 * b] / (double)host[2 * b + 1]);
  if (r.empty() || wall_khz <= 0) return hipErrorUnknown;
  std::nth_element(r.begin(), r.begin() + r.size() / 2, r.end());
  *mhz = r[r.size() / 2] * (double)wall_khz * 1e-3;
  return hipSuccess;
}

namespace {
}  // namespace

#ifdef NUFFT_HIP_PHASE_LOG
extern "C" int nufft_hip_debug_phase_log3(unsigned long long* dst, int n) {   // experiment build only
  if (n > kPhaseLogWgs3 * kPhaseSlots3) n = kPhaseLogWgs3 * kPhaseSlots3;
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_phase_log3), sizeof(unsigned long long) * (size_t)n);
}
#endif

bool patch3_supported(const Geom& g, int precision) {
  return precision == NUFFT_HIP_F32 && g.rank == 3 && g.fx_patch && (g.w == 7 || g.w == 8) && g.tile[0] == kDenseTile &&
         g.tile[1] == kDenseTile && g.tile[2] == 8;
}
constexpr int kPatchHalf = 8;
size_t patch3_lds_bytes(int w) {
  return w == 8 ? PatchCfg<8, 8, kPatchHalf>::lds_bytes : w == 7 ? PatchCfg<7, 8, kPatchHalf>::lds_bytes : 0;
}
template <int W>
static hipError_t launch_patch3(const Geom& g, const SortedPoints<float>& sp, const float* horner, const float* c, float* fw,
                                dim3 grid, int64_t c_stride, int64_t fw_stride, float scale, hipStream_t stream) {
  using C = PatchCfg<W, 8, kPatchHalf>;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(spread_patch3_kernel<W, 8, kPatchHalf>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes);
  if (e != hipSuccess) return e;
  spread_patch3_kernel<W, 8, kPatchHalf><<<grid, kPatchNW * 64, C::lds_bytes, stream>>>(g, sp, horner, c, fw, c_stride, fw_stride, scale);
  return hipGetLastError();
}
hipError_t launch_spread_group3_fallback(const Geom& g, const SortedPoints<float>& sp, unsigned nsub_bound, const float* horner,
                                         const float* c, float* fw, int batch, int64_t c_stride, int64_t fw_stride, float scale,
                                         hipStream_t stream) {
  if (!sp.fb_list || !sp.fb_ticket) return hipErrorInvalidValue;
  const hipError_t e0 = hipMemsetAsync(sp.fb_ticket, 0, sizeof(int), stream);
  if (e0 != hipSuccess) return e0;
  // (r05 advisor) ~150 KB of LDS: one workgroup per CU, so a grid beyond the CU count only queues on the ticket -- fixed latency
  // on small transforms whose list is empty, the common case
  static int ncu = 0;
  if (ncu == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    ncu = v;
  }
  const dim3 grid((unsigned)std::min<uint64_t>((uint64_t)nsub_bound * (unsigned)batch, (uint64_t)ncu), 1u);
#define NUFFT_FB(WV)                                                                                                          \
  {                                                                                                                           \
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(spread_group3_f64_kernel<WV>),                     \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)FbCfg<WV>::lds_bytes);          \
    if (e != hipSuccess) { (void)hipGetLastError(); return hipErrorNotSupported; }   /* a part with less LDS: the caller takes the per-point launches */ \
    spread_group3_f64_kernel<WV><<<grid, kFbNW * 64, FbCfg<WV>::lds_bytes, stream>>>(g, sp, horner, c, fw, c_stride, fw_stride, scale, batch); \
  }
  switch (g.w) {
    case 8: NUFFT_FB(8) break;
    case 7: NUFFT_FB(7) break;
    case 6: NUFFT_FB(6) break;
    case 5: NUFFT_FB(5) break;
    case 4: NUFFT_FB(4) break;
    case 3: NUFFT_FB(3) break;
    case 2: NUFFT_FB(2) break;
    default: return hipErrorInvalidValue;
  }
#undef NUFFT_FB
  return hipGetLastError();
}

hipError_t launch_spread_patch3(const Geom& g, const SortedPoints<float>& sp, unsigned nsub_bound, const float* horner,
                                const float* c, float* fw, int batch, int64_t c_stride, int64_t fw_stride, float scale,
                                hipStream_t stream) {
  const dim3 grid(nsub_bound, (unsigned)batch);
  if (g.w == 8) return launch_patch3<8>(g, sp, horner, c, fw, grid, c_stride, fw_stride, scale, stream);
  if (g.w == 7) return launch_patch3<7>(g, sp, horner, c, fw, grid, c_stride, fw_stride, scale, stream);
  return hipErrorInvalidValue;
}
hipError_t launch_bound3(const Geom& g, const Rec<float>* rec, int rec_stride, const int32_t* tile_start, const int32_t* sub_start,
                         unsigned nsub_bound, const TapMax& taps, float* sub_bound, int* fb_list, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(fb_list, 0, sizeof(int), stream);
  if (e != hipSuccess) return e;
  if (nsub_bound == 0) return hipSuccess;
  if (g.w == 8) bound3_kernel<8, 8><<<nsub_bound, kBoundThreads, 0, stream>>>(g, rec, rec_stride, tile_start, sub_start, taps, sub_bound, fb_list);
  else if (g.w == 7) bound3_kernel<7, 8><<<nsub_bound, kBoundThreads, 0, stream>>>(g, rec, rec_stride, tile_start, sub_start, taps, sub_bound, fb_list);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
// ---- stacks (r05) ------------------------------------------------------------------------------------------------
// Points per stack and tiles per stack. Tiles: a stack of nz tiles writes 8 + (W - 1) / nz planes per tile, and
// 512 stacks (one per workgroup slot of the chip) already beat 2048 shorter ones (r05 sweep, 128^3 modes at 0.05
// points per cell: 16 tiles 313 us, 8 320, 4 334, 2 372; 256 stacks of 32 tiles 399): at least 512 stacks where the
// tile count allows, at most 16 tiles. Points: 8192, so that workgroups stay of similar length whatever the density
// (and never below the subproblem cap: pieces are subproblems).
constexpr double kStackDensity = 0.22;   // points per fine cell below which a w = 7, 8 plan spreads over stacks
constexpr double kStack64Density = 2.0;  // the same for double-precision plans (r06)
void stack_params(const Geom& g, int* cap, int* len) {
  int l = (g.ntile[0] * g.ntile[1] * g.ntile[2] * (g.nitems > 1 ? g.nitems : 1)) / 512;   // (the tiles that exist: ntiles counts the ids, padded under the super-tile numbering)
  l = l < 2 ? 2 : (l > 16 ? 16 : l);
  if (l > g.ntile[2]) l = g.ntile[2];
  *len = g.stack_len > 0 ? g.stack_len : l;
  *cap = std::max(g.stack_cap > 0 ? g.stack_cap : 8192, g.max_sub);
}
// launch grid of the stack kernels: an upper bound on the stacks stack_plan_kernel can cut (known without reading
// the device): per column ceil(ntz / len) closed by the length rule + the open one; a stack closed by the point cap
// holds more than cap points together with its successor; a tile above max_sub points closes one stack and becomes
// at most n / max_sub + 1 pieces
unsigned stack_grid_bound(const Geom& g, int64_t M) {
  int cap, len;
  stack_params(g, &cap, &len);
  const int64_t ncol = (int64_t)g.ntile[0] * g.ntile[1] * std::max(1, g.nitems);
  const int64_t n = ncol * ((g.ntile[2] + len - 1) / len + 1) + 2 * M / cap + 3 * M / g.max_sub + 2;
  return (unsigned)std::min<int64_t>(n, 0x7fffffff);
}
// The stack form pays where the tile + halo write-out is a visible part of the spread (r05, profiles/r05_stack_ab.txt,
// spread stage, stacks against subproblems, w = 8): 256^3 modes at 0.022 points per fine cell 2.00 against 2.98 ms,
// 0.075: 2.38 against 3.60, 0.149: 3.35 against 4.65, 0.179: 3.84 against 4.15, 0.201: 4.17 against 4.40, 0.224: 4.49 against
// 4.58 and 0.75: 12.7 against 12.4 -- from there the accumulation loop (VALU issue + LDS atomics) hides the write-out
// either way, and the bound of a stack costs 0.1 ms more than the bounds of its tiles.
// options.tuning STACK_OFF / STACK_ON force the choice.
bool stack3_wanted(const Geom& g, int64_t M) {
  // (w = 7, 8: spread_stack3_kernel; w <= 6 on 16 x 16 x 8 tiles: spread_dense3_stack_kernel)
  const bool dense = g.fixed_point && !g.fx_patch && g.rank == 3 && g.w >= 2 && g.w <= 6 && g.tile[0] == kDenseTile &&
                     g.tile[1] == kDenseTile && g.tile[2] == 8;
  if ((!g.fx_patch && !dense && !g.fp64_stack) || g.ntile[2] < 2 || g.ntile[2] > 32767) return false;
  const int mode = tune_mode(g, NUFFT_HIP_TUNE_STACK_OFF, NUFFT_HIP_TUNE_STACK_ON);
  if (mode >= 0) return mode != 0;
  const double cells = (double)g.nf[0] * g.nf[1] * g.nf[2] * (g.nitems > 1 ? g.nitems : 1);
  // double precision (spread_wave3_stack_kernel, nufft_kernels.hip): the depth-4 tiles write 5.7 x the fine grid per
  // subproblem, and a workgroup per ~1000 fine cells zeroes 97 KB of planes first
  // (w = 9..16, spread_wide_kernel<..., STACK>: 256^3 modes, tol 1e-9, 0.075 points per fine cell 37.6 -> 17.3 ms, 0.22: 41.4 -> 35.2;
  // 128^3 modes, tol 1e-12, 0.6 and 1.8 per cell: 17.4 -> 17.7, 47.7 -> 48.8: profiles/r06_c128_widestack.txt)
  if (g.fp64_stack) return (double)M < (g.wide ? 0.5 : kStack64Density) * cells;
  // w <= 6 (spread_dense3_stack_kernel, same file): the fewer atomics a point costs, the longer the write-out shows --
  // w = 6 (4 per point), 256^3 modes, spread stage: 0.075 per cell 2.93 -> 2.33 ms, 0.224: 4.06 -> 3.33, 0.298: 3.67 -> 3.84,
  // 0.745: 7.55 -> 7.91; w = 4 (1 per point): 0.224: 2.87 -> 2.33, 0.373: 3.00 -> 2.81, 0.745: 4.50 -> 4.33
  const double limit = g.fx_patch ? kStackDensity : (g.w >= 5 ? 0.25 : 1.0);
  return (double)M < limit * cells;
}
hipError_t launch_stack_plan(const Geom& g, const int32_t* tile_start, int64_t M, int4* segs, int* seg_count, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(seg_count, 0, sizeof(int), stream);
  if (e != hipSuccess) return e;
  int cap, len;
  stack_params(g, &cap, &len);
  const int ncol = g.ntile[0] * g.ntile[1] * std::max(1, g.nitems);
  stack_plan_kernel<<<(unsigned)((ncol + kStackPlanWaves - 1) / kStackPlanWaves), kStackPlanWaves * 64, 0, stream>>>(
      g, tile_start, ncol, cap, len, segs, seg_count, (int)stack_grid_bound(g, M));
  return hipGetLastError();
}
hipError_t launch_bound3_stack(const Geom& g, const Rec<float>* rec, int rec_stride, const int32_t* tile_start,
                               const int32_t* sub_start, int64_t M, const TapMax& taps, const int4* segs, const int* seg_count,
                               float* seg_bound, int* fb_list, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(fb_list, 0, sizeof(int), stream);
  if (e != hipSuccess) return e;
  const unsigned grid = stack_grid_bound(g, M);
  if (g.w == 8) bound3_stack_kernel<8, 8><<<grid, kBoundThreads, 0, stream>>>(g, rec, rec_stride, tile_start, sub_start, taps, segs, seg_count, seg_bound, fb_list);
  else if (g.w == 7) bound3_stack_kernel<7, 8><<<grid, kBoundThreads, 0, stream>>>(g, rec, rec_stride, tile_start, sub_start, taps, segs, seg_count, seg_bound, fb_list);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
template <int W>
static hipError_t launch_stack3(const Geom& g, const SortedPoints<float>& sp, const float* horner, const float* c, float* fw,
                                dim3 grid, int64_t c_stride, int64_t fw_stride, float scale, hipStream_t stream) {
  using C = PatchCfg<W, 8, kPatchHalf>;
  const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(spread_stack3_kernel<W, 8, kPatchHalf>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::lds_bytes);
  if (e != hipSuccess) return e;
  spread_stack3_kernel<W, 8, kPatchHalf><<<grid, kPatchNW * 64, C::lds_bytes, stream>>>(g, sp, horner, c, fw, c_stride, fw_stride, scale);
  return hipGetLastError();
}
// the w <= 6 kernel over stacks
template <int W, bool FUSED>
static hipError_t launch_dense3_stack(const Geom& g, const SortedPoints<float>& sp, const float* horner, const float* c, float* fw,
                                      dim3 grid, int64_t c_stride, int64_t fw_stride, float scale, hipStream_t stream) {
  using C = DenseCfg<W, 8>;
  constexpr size_t lds = C::lds_bytes;
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(spread_dense3_stack_kernel<W, 8, FUSED>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  spread_dense3_stack_kernel<W, 8, FUSED><<<grid, C::NW * 64, lds, stream>>>(g, sp, horner, c, fw, c_stride, fw_stride, scale);
  return hipGetLastError();
}
hipError_t launch_spread_dense3_stack(const Geom& g, const SortedPoints<float>& sp, int64_t M, const float* horner,
                                      const float* c, float* fw, int batch, int64_t c_stride, int64_t fw_stride, float scale,
                                      hipStream_t stream) {
  const dim3 grid(stack_grid_bound(g, M), (unsigned)batch);
#define NUFFT_D3S(WV)                                                                                                 \
  case WV:                                                                                                            \
    return g.fused ? launch_dense3_stack<WV, true>(g, sp, horner, c, fw, grid, c_stride, fw_stride, scale, stream)    \
                   : launch_dense3_stack<WV, false>(g, sp, horner, c, fw, grid, c_stride, fw_stride, scale, stream);
  switch (g.w) {
    NUFFT_D3S(2) NUFFT_D3S(3) NUFFT_D3S(4) NUFFT_D3S(5) NUFFT_D3S(6)
    default: return hipErrorInvalidValue;
  }
#undef NUFFT_D3S
}
hipError_t launch_spread_stack3(const Geom& g, const SortedPoints<float>& sp, int64_t M, const float* horner,
                                const float* c, float* fw, int batch, int64_t c_stride, int64_t fw_stride, float scale,
                                hipStream_t stream) {
  const dim3 grid(stack_grid_bound(g, M), (unsigned)batch);
  if (g.w == 8) return launch_stack3<8>(g, sp, horner, c, fw, grid, c_stride, fw_stride, scale, stream);
  if (g.w == 7) return launch_stack3<7>(g, sp, horner, c, fw, grid, c_stride, fw_stride, scale, stream);
  return hipErrorInvalidValue;
}
int cstats_blocks(int64_t M, int slots) {
  // (workgroups per slot: 16384 strengths each, at most 1024, and at most 2^20 partial pairs over all slots)
  int64_t n = std::min<int64_t>(kStatsMaxBlocks, (M + 16383) / 16384);
  n = std::min<int64_t>(n, ((int64_t)1 << 20) / std::max(1, slots));
  return (int)std::max<int64_t>(1, n);
}
size_t cstats_floats(int64_t M, int slots) { return 2 * (size_t)slots * (1 + (size_t)cstats_blocks(M, slots)); }
// cstats: [max_slots][2] results followed by the partial pairs (cstats_floats(M, max_slots) floats in all)
hipError_t launch_cstats(const float* c, int64_t M, int slots, int nblk, int max_slots, int64_t c_stride, float* cstats, hipStream_t stream) {
  if (slots <= 0) return hipSuccess;
  if (M <= 0 || nblk <= 0) return hipMemsetAsync(cstats, 0, sizeof(float) * 2 * (size_t)slots, stream);
  float* partial = cstats + 2 * (size_t)max_slots;
  cstats_partial_kernel<<<dim3((unsigned)nblk, (unsigned)slots), 256, 0, stream>>>(reinterpret_cast<const float2*>(c), M, c_stride, partial);
  cstats_final_kernel<<<(unsigned)slots, 256, 0, stream>>>(partial, nblk, cstats);
  return hipGetLastError();
}

bool dense3_supported(const Geom& g, int precision) {
  // (tiles of depth 4 -- only on request, options.tile_dims -- stay on spread_wave3_kernel)
  return precision == NUFFT_HIP_F32 && g.rank == 3 && g.fixed_point && g.w >= 2 && g.w <= 6 && g.tile[0] == kDenseTile &&
         g.tile[1] == kDenseTile && g.tile[2] == 8;
}

size_t dense3_lds_bytes(int w) {
#define NUFFT_D3(WV) case WV: return DenseCfg<WV, 8>::lds_bytes_group;
  switch (w) {
    NUFFT_D3(2) NUFFT_D3(3) NUFFT_D3(4) NUFFT_D3(5) NUFFT_D3(6)
    default: return 0;
  }
#undef NUFFT_D3
}

// When the cell-grouped form pays (r03, w = 6, spread stage, grouped against not): runs per point are
// (1 - exp(-d)) / d at d points per fine cell -- 0.70 at config 4's 0.75, where the in-LDS sort and the register adds
// cost what the saved atomics gain (fused records 7.63-7.77 against 7.73-7.81 ms; unfused 8.69 against 8.14: the
// records AND the strengths are then gathered through the permutation); 0.46 at 1.8 points per cell: 1.95 against
// 2.23 ms (128^3, M = 3e7). At w <= 4 a point is ONE atomic and there is nothing to save (5.6 against 5.0 ms): not
// instantiated. Taken from 1.2 points per cell; options.tuning GROUP_OFF / GROUP_ON force the choice (w >= 5).
bool dense3_grouped(const Geom& g, int64_t M) {
  if (g.w < 5 || g.max_sub > DenseCfg<6, 8>::kMaxSub) return false;
  const int mode = tune_mode(g, NUFFT_HIP_TUNE_GROUP_OFF, NUFFT_HIP_TUNE_GROUP_ON);
  if (mode >= 0) return mode != 0;
  return (double)M >= 1.2 * (double)g.nf[0] * (double)g.nf[1] * (double)g.nf[2];
}

hipError_t launch_spread_dense3(const Geom& g, const SortedPoints<float>& sp, unsigned nsub_bound, int64_t M, const float* horner,
                                const float* c, float* fw, int batch, int64_t c_stride, int64_t fw_stride, float scale,
                                hipStream_t stream) {
  const dim3 grid(nsub_bound, (unsigned)batch);
  const bool grouped = dense3_grouped(g, M);
#define NUFFT_D3(WV)                                                                                                  \
  case WV:                                                                                                            \
    if constexpr (WV >= 5) {                                                                                          \
      if (grouped)                                                                                                    \
        return g.fused ? launch_dense3<WV, 8, true, true>(g, sp, horner, c, fw, grid, c_stride, fw_stride, scale, stream)   \
                       : launch_dense3<WV, 8, false, true>(g, sp, horner, c, fw, grid, c_stride, fw_stride, scale, stream); \
    }                                                                                                                 \
    return g.fused ? launch_dense3<WV, 8, true, false>(g, sp, horner, c, fw, grid, c_stride, fw_stride, scale, stream)    \
                   : launch_dense3<WV, 8, false, false>(g, sp, horner, c, fw, grid, c_stride, fw_stride, scale, stream);
  switch (g.w) {
    NUFFT_D3(2) NUFFT_D3(3) NUFFT_D3(4) NUFFT_D3(5) NUFFT_D3(6)
    default: return hipErrorInvalidValue;
  }
#undef NUFFT_D3
}

}  // namespace nufft_hip
