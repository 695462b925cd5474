// gfx950 interpolation kernel for 1-D plans (reference InterpSubproblem1DKernel /
// InterpNuptsDriven1DKernel, nufft_plan.cu.cc:653-704): replaces the thread-per-point kernel of
// nufft_kernels.hip that gathers straight from global memory. The tile and its halo are loaded
// into LDS, one thread per point evaluates its W kernel values in registers and reads its W
// consecutive cells (1-D fp64 tol 1e-9, M = 1e7: 0.96 -> 0.71 ms per transform; float 0.57 -> 0.53;
// profiles/r02_line_interp_ab.txt).
//
// 1-D SPREADING stays on the thread-per-point tile kernel (spread_tile_generic_kernel). A
// cell-owner gather kernel was built and measured against it (r02): every subproblem
// counting-sorted by start cell in LDS, one thread per output cell summing, for each stencil
// offset, the points that start there -- no accumulation atomics. At a few points per cell the
// two were level (226 against 234 us at M = 1e7, 2^21 fine cells), with the run-time-length
// polynomials of w > 8 the gather lost (fp64 tol 1e-9: 1.34 against 0.76 ms), and it only won
// on point sets a thousand times denser than the grid (232 against 345 us), so it was removed.
#include <cstdio>
#include <cstdlib>

#include "nufft_hip_internal.h"
#include "nufft_device.h"

namespace nufft_hip {

namespace {

constexpr int kLineFixedCoef = 10;    // every width <= 8 fits (rows above the fitted count are zero)

template <typename T, int W>
__device__ __forceinline__ void line_horner(const T* __restrict__ tab, int nc, T z, T (&k)[W]) {
  if (nc <= kLineFixedCoef) {
#pragma unroll
    for (int q = 0; q < W; ++q) k[q] = tab[(kLineFixedCoef - 1) * kMaxW + q];
#pragma unroll
    for (int t = kLineFixedCoef - 2; t >= 0; --t) {
#pragma unroll
      for (int q = 0; q < W; ++q) k[q] = fma_sgpr(k[q], z, tab[t * kMaxW + q]);
    }
  } else {
#pragma unroll
    for (int q = 0; q < W; ++q) k[q] = tab[(nc - 1) * kMaxW + q];
    for (int t = nc - 2; t >= 0; --t) {
#pragma unroll
      for (int q = 0; q < W; ++q) k[q] = fma_sgpr(k[q], z, tab[t * kMaxW + q]);
    }
  }
}

constexpr int kLineInterpThreads = 256;
template <typename T, int W>
__global__ __launch_bounds__(kLineInterpThreads) void interp_line_kernel(
    Geom g, SortedPoints<T> sp, const T* __restrict__ horner, T* __restrict__ c,
    const T* __restrict__ fw, int64_t c_stride, int64_t fw_stride, T scale) {
  using T2 = typename Pair<T>::type;
  constexpr int NT = kLineInterpThreads;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  T2* tile = reinterpret_cast<T2*>(smem_raw);
  int tb, p0, p1, slot;
  if (!locate_subproblem(g, sp.tile_start, sp.sub_start, blockIdx.x, &tb, &p0, &p1, &slot)) return;
  const int tid = threadIdx.x;
  const int tile0 = g.tile[0];
  const int L0 = tile0 + W - 1;
  const int o0 = tb * tile0;
  const T2* in = reinterpret_cast<const T2*>(fw) + (int64_t)slot * fw_stride;
  for (int i = tid; i < L0; i += NT) tile[i] = in[wrap1(o0 + i, g.nf[0])];
  __syncthreads();
  const int nc = g.ncoef;
  T2* cc = reinterpret_cast<T2*>(c) + (int64_t)slot * c_stride;
  // the next record is requested (on a clamped index) before this point's arithmetic
  Rec<T> raw = sp.rec[p0 + tid < p1 ? p0 + tid : p1 - 1];
  for (int j = p0 + tid; j < p1; j += NT) {
    const PointView<T> rec = unpack_rec<T, 1>(raw);
    raw = sp.rec[j + NT < p1 ? j + NT : p1 - 1];
    T k[W];
    line_horner<T, W>(horner, nc, rec.z0, k);
    const T2* tp = tile + (int)(rec.loc & 1023);
    T sre = (T)0, sim = (T)0;
#pragma unroll
    for (int q = 0; q < W; ++q) {
      const T2 v = tp[q];
      sre = fma(k[q], v.x, sre);
      sim = fma(k[q], v.y, sim);
    }
    T2 out;
    out.x = sre * scale;
    out.y = sim * scale;
    cc[rec.idx] = out;
  }
}

}  // namespace

// Tiles of at most 1024 cells (the 10-bit tile-local start of the records).
bool line_kernels_supported(const Geom& g) { return g.rank == 1 && g.w >= 2 && g.w <= 16 && g.tile[0] <= 1024; }
size_t line_interp_lds_bytes(const Geom& g, int precision) { return (size_t)(g.tile[0] + g.w - 1) * 2 * precision; }

template <typename T>
hipError_t launch_interp_line(const Geom& g, const SortedPoints<T>& sp, int64_t M, const T* horner, T* c,
                              const T* fw, int batch, int64_t c_stride, int64_t fw_stride, T scale,
                              hipStream_t stream) {
  if (M == 0) return hipSuccess;
  dim3 grid((unsigned)((int64_t)g.ntiles + M / g.max_sub), (unsigned)batch);
  const size_t lds = line_interp_lds_bytes(g, (int)sizeof(T));
#define NUFFT_LINE_CASE(WW)                                                                              \
  case WW:                                                                                               \
    interp_line_kernel<T, WW><<<grid, kLineInterpThreads, lds, stream>>>(g, sp, horner, c, fw, c_stride, \
                                                                         fw_stride, scale);              \
    break;
  switch (g.w) {
    NUFFT_LINE_CASE(2) NUFFT_LINE_CASE(3) NUFFT_LINE_CASE(4) NUFFT_LINE_CASE(5) NUFFT_LINE_CASE(6)
    NUFFT_LINE_CASE(7) NUFFT_LINE_CASE(8) NUFFT_LINE_CASE(9) NUFFT_LINE_CASE(10) NUFFT_LINE_CASE(11)
    NUFFT_LINE_CASE(12) NUFFT_LINE_CASE(13) NUFFT_LINE_CASE(14) NUFFT_LINE_CASE(15) NUFFT_LINE_CASE(16)
    default: return hipErrorInvalidValue;
  }
#undef NUFFT_LINE_CASE
  return hipGetLastError();
}
template hipError_t launch_interp_line<float>(const Geom&, const SortedPoints<float>&, int64_t, const float*,
                                              float*, const float*, int, int64_t, int64_t, float, hipStream_t);
template hipError_t launch_interp_line<double>(const Geom&, const SortedPoints<double>&, int64_t, const double*,
                                               double*, const double*, int, int64_t, int64_t, double,
                                               hipStream_t);

}  // namespace nufft_hip
