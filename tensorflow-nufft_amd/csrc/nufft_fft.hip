// Pruned FFT passes of the NUFFT plan (gfx950): fft_rotate_kernel for power-of-two fine-grid dimensions up to 2048,
// fft_mixed_kernel (r06, second half of this file) for every other even 2^a 3^b 5^c length up to 4096.
//
// The reference runs a full in-place FFT of the oversampled grid (cuFFT / FFTW,
// nufft_plan.cu.cc:2147-2152, nufft_plan.cc:336) and then a separate deconvolve kernel
// that keeps only the N of the nf = sigma N modes per dimension
// (Deconvolve*/Amplify*, nufft_plan.cu.cc:326-435). Only those N modes are ever used, so
// here the transform is done one dimension at a time and every pass already crops (type 1)
// or zero-pads (type 2) its dimension and applies that dimension's deconvolution factor:
//
//   type 1:  fine[z][y][x] --x--> B1[kx][z][y] --y--> B2[ky][kx][z] --z--> f[kz][ky][kx]
//   type 2:  f[kz][ky][kx] --x--> B1[x][kz][ky] --y--> B2[y][x][kz] --z--> fine[z][y][x]
//
// Every pass reads contiguous lines, transforms them in LDS and writes its output
// TRANSPOSED ([bin][line]), which makes the next dimension contiguous and, after `rank`
// passes, restores the original axis order. Type 1 at sigma = 2 moves 1 + 1/2 (+ 1/4 ...)
// grids instead of the 8 the rocFFT transpose pipeline moves for a 2-D grid (4 kernels,
// each reading and writing the whole grid) plus the deconvolve pass: 2048^2 complex64,
// 72 us (rocFFT) + 9 us (deconvolve) -> see DESIGN.md for the measured figure.
//
// One workgroup of 256 threads transforms R lines (n / 8 threads per line, 2048 / n lines at
// a time): Stockham autosort passes of radix 8 / 4 / 2 with the butterflies in registers and
// the exchange through an LDS line buffer; the results collect in an LDS tile [bins][R] that
// is written out with R consecutive lines per bin (64-128 byte segments).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "nufft_hip_internal.h"

namespace nufft_hip {

namespace {

template <typename T> struct C2;
template <> struct C2<float> { using type = float2; };
template <> struct C2<double> { using type = double2; };

template <typename V, typename T>
__device__ __forceinline__ V cmul(V a, V b) {
  V r;
  r.x = a.x * b.x - a.y * b.y;
  r.y = a.x * b.y + a.y * b.x;
  return r;
}
template <typename V> __device__ __forceinline__ V cadd(V a, V b) { V r; r.x = a.x + b.x; r.y = a.y + b.y; return r; }
template <typename V> __device__ __forceinline__ V csub(V a, V b) { V r; r.x = a.x - b.x; r.y = a.y - b.y; return r; }
// multiply by sgn * i  (sgn = -1: forward transform, exp(-i ...))
template <typename V, typename T>
__device__ __forceinline__ V mul_i(V a, T sgn) { V r; r.x = -sgn * a.y; r.y = sgn * a.x; return r; }

template <typename V, typename T>
__device__ __forceinline__ void dft2(V& a, V& b) {
  const V t = csub(a, b);
  a = cadd(a, b);
  b = t;
}
// in-place DFT of size 4, natural output order
template <typename V, typename T>
__device__ __forceinline__ void dft4(V& v0, V& v1, V& v2, V& v3, T sgn) {
  const V a = cadd(v0, v2), b = csub(v0, v2), c = cadd(v1, v3), d = mul_i<V, T>(csub(v1, v3), sgn);
  v0 = cadd(a, c);
  v1 = cadd(b, d);
  v2 = csub(a, c);
  v3 = csub(b, d);
}
template <typename V, typename T>
__device__ __forceinline__ void dft8(V (&v)[8], T sgn) {
  V e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
  V o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
  dft4<V, T>(e0, e1, e2, e3, sgn);
  dft4<V, T>(o0, o1, o2, o3, sgn);
  const T h = (T)0.70710678118654752440;
  // w8^t = exp(sgn i pi t / 4)
  V w1; w1.x = h; w1.y = sgn * h;
  V w3; w3.x = -h; w3.y = sgn * h;
  o1 = cmul<V, T>(o1, w1);
  o2 = mul_i<V, T>(o2, sgn);
  o3 = cmul<V, T>(o3, w3);
  v[0] = cadd(e0, o0); v[4] = csub(e0, o0);
  v[1] = cadd(e1, o1); v[5] = csub(e1, o1);
  v[2] = cadd(e2, o2); v[6] = csub(e2, o2);
  v[3] = cadd(e3, o3); v[7] = csub(e3, o3);
}

constexpr int kFftMaxThreads = 1024;

template <typename T>
struct FftPassArgs {
  const typename C2<T>::type* in;
  typename C2<T>::type* out;
  const typename C2<T>::type* tw;   // exp(sgn 2 pi i m / n), m = 0 .. n-1
  const T* rf;                      // reciprocal kernel Fourier series of this dimension, [n / 2 + 1]
  int n, kin, kout;                 // FFT length; input / output line lengths (n: all bins; < n: CMCL modes)
  int64_t nlines;                   // lines per transform
  int64_t in_batch, out_batch;      // elements between consecutive transforms
  int R;                            // lines per workgroup
  int LW;                           // lines in flight (blockDim.x = LW * n / 8)
  int zero_in;                      // store zeros over every input element after reading it (type 1, first
                                    // dimension: leaves the fine grid cleared for the next spread)
  int npass;
  unsigned radpack;                 // radix of pass p in bits [4 p, 4 p + 4) (an indexed array in the kernel
                                    // arguments makes the compiler copy them to scratch memory)
  float sgn;
};

// bin m of an n-point transform <-> mode k in [-(K/2), (K-1)/2] stored at index k + K/2
// (CMCL order); returns -1 for bins outside the kept modes
__device__ __forceinline__ int bin_to_mode_index(int m, int n, int K, int* absk) {
  int k;
  if (m <= (K - 1) / 2) k = m;
  else if (m >= n - K / 2) k = m - n;
  else return -1;
  *absk = k < 0 ? -k : k;
  return k + K / 2;
}

// Line buffer index with one pad element per 32: the first exchange writes with a stride of
// `radix` elements between consecutive lanes (16-way bank conflicts on 8-byte elements
// without the pad, none with it); reads are always consecutive.
__device__ __forceinline__ int lpad(int i) { return i + (i >> 5); }
__host__ __device__ __forceinline__ int lpad_len(int n) { return n + (n >> 5) + 1; }

// Per-thread maps between the 8 bins a thread touches in the first / last pass and the CMCL
// mode array of a line (type 2 input: zero-padded; type 1 output: cropped), with the
// deconvolution factor of each. Computed once per workgroup: they do not depend on the line.
template <typename T>
struct BinMap { int idx[8]; T sc[8]; };

// First-pass inputs of one line, straight from global memory. Butterfly u of the thread takes
// elements lt + u TL + t n / RAD, t < RAD. PAD: the line holds kin < n modes (see BinMap).
template <typename T, int RAD, bool PAD>
__device__ __forceinline__ void fft_load_line(const typename C2<T>::type* rowp, bool live, int lt,
                                              int TL, int n, const BinMap<T>& map, typename C2<T>::type (&x)[8],
                                              bool zero_in = false) {
  using V = typename C2<T>::type;
  const int stride = n / RAD;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int u = q / RAD, t = q % RAD;
    V y; y.x = (T)0; y.y = (T)0;
    if constexpr (PAD) {
      const V z = rowp[map.idx[q] >= 0 ? map.idx[q] : 0];   // unconditional loads
      if (live && map.idx[q] >= 0) { y.x = z.x * map.sc[q]; y.y = z.y * map.sc[q]; }
    } else {
      const V z = rowp[lt + u * TL + t * stride];
      if (live) y = z;
    }
    x[q] = y;
  }
  if constexpr (!PAD) {
    if (zero_in && live) {   // (same thread, same addresses: ordered behind the loads)
      V zero; zero.x = (T)0; zero.y = (T)0;
      V* w = const_cast<V*>(rowp);
#pragma unroll
      for (int q = 0; q < 8; ++q) w[lt + (q / RAD) * TL + (q % RAD) * stride] = zero;
    }
  }
}

// One Stockham pass of radix RAD on the 8 values of a thread (8 / RAD butterflies).
template <typename T, int RAD, bool CROP, bool INPLACE = false>
__device__ __forceinline__ void fft_pass(const typename C2<T>::type* twp, int n, const BinMap<T>& map,
                                         typename C2<T>::type (&v)[8], typename C2<T>::type* mybuf,
                                         typename C2<T>::type* trow, int lt, int TL, int Ns, bool first, bool last,
                                         bool row_ok, T sgn) {
  using V = typename C2<T>::type;
  const int stride = n / RAD;
  if (!first) {
    // inputs from the line buffer, then the twiddles exp(sgn 2 pi i t k / (Ns RAD)), k = j mod Ns
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int u = q / RAD, t = q % RAD;
      v[q] = mybuf[lpad(lt + u * TL + t * stride)];
    }
    const int tstep = n / (Ns * RAD);
#pragma unroll
    for (int u = 0; u < 8 / RAD; ++u) {
      const int k = (lt + u * TL) & (Ns - 1);
      const int base = k * tstep;
#pragma unroll
      for (int t = 1; t < RAD; ++t) {
        // half table in LDS: exp(i pi) = -1 gives the other half
        const int ti = (t * base) & (n - 1);
        V w = twp[ti & (n / 2 - 1)];
        if (ti & (n / 2)) { w.x = -w.x; w.y = -w.y; }
        v[u * RAD + t] = cmul<V, T>(v[u * RAD + t], w);
      }
    }
  }
  if constexpr (RAD == 8) {
    dft8<V, T>(v, sgn);
  } else if constexpr (RAD == 4) {
    dft4<V, T>(v[0], v[1], v[2], v[3], sgn);
    dft4<V, T>(v[4], v[5], v[6], v[7], sgn);
  } else {
    dft2<V, T>(v[0], v[1]); dft2<V, T>(v[2], v[3]); dft2<V, T>(v[4], v[5]); dft2<V, T>(v[6], v[7]);
  }
  if (!last) {
    __syncthreads();   // every thread has read its inputs of this pass
#pragma unroll
    for (int u = 0; u < 8 / RAD; ++u) {
      const int j = lt + u * TL;
      const int k = j & (Ns - 1);
      const int o = (j - k) * RAD + k;
#pragma unroll
      for (int t = 0; t < RAD; ++t) mybuf[lpad(o + t * Ns)] = v[u * RAD + t];
    }
    __syncthreads();
  } else if (row_ok) {
    // natural-order bins -> tile [line][bin]; CROP: only the kept modes, deconvolved (map)
#pragma unroll
    for (int u = 0; u < 8 / RAD; ++u) {
      const int j = lt + u * TL;
      const int k = j & (Ns - 1);
      const int o = (j - k) * RAD + k;
#pragma unroll
      for (int t = 0; t < RAD; ++t) {
        const V val = v[u * RAD + t];
        if constexpr (CROP) {
          const int idx = map.idx[u * RAD + t];
          if (idx >= 0) {
            const T sc = map.sc[u * RAD + t];
            V y; y.x = val.x * sc; y.y = val.y * sc;
            trow[idx] = y;
          }
        } else if constexpr (INPLACE) {
          trow[lpad(o + t * Ns)] = val;   // (the output row is the line buffer itself: padded layout)
        } else {
          trow[o + t * Ns] = val;
        }
      }
    }
  }
}

// Radix of pass P of a 2^LOGN-point transform: radix 8 while three bits remain, then 4 or 2.
template <int LOGN, int P> constexpr int fft_radix() {
  constexpr int done = 3 * P;
  constexpr int rem = LOGN - done;
  return rem >= 3 ? 8 : rem == 2 ? 4 : rem == 1 ? 2 : 1;
}
template <int LOGN> constexpr int fft_npass() { return (LOGN + 2) / 3; }

template <typename T, int LOGN, int P, bool CROP, bool INPLACE = false>
__device__ __forceinline__ void fft_all_passes(const typename C2<T>::type* twp, const BinMap<T>& map,
                                               typename C2<T>::type (&v)[8], typename C2<T>::type* mybuf,
                                               typename C2<T>::type* trow, int lt, bool row_ok, T sgn) {
  constexpr int NP = fft_npass<LOGN>();
  if constexpr (P < NP) {
    constexpr int RAD = fft_radix<LOGN, P>();
    constexpr int Ns = 1 << (3 * P);   // every earlier pass has radix 8
    fft_pass<T, RAD, CROP, INPLACE>(twp, 1 << LOGN, map, v, mybuf, trow, lt, (1 << LOGN) / 8, Ns, P == 0, P == NP - 1, row_ok, sgn);
    fft_all_passes<T, LOGN, P + 1, CROP, INPLACE>(twp, map, v, mybuf, trow, lt, row_ok, sgn);
  }
}

// PAD = false: type-1 pass (plain input lines of n points, output cropped to kout modes);
// PAD = true: type-2 pass (input lines of kin modes zero-padded to n, all n bins written).
// GATHER (type 2): the mirror image of the data flow. The input holds the workgroup's R lines
// INTERLEAVED (element m of line l at in[m nlines + l]: R x 8-byte runs, gathered into the LDS tile
// by the whole workgroup) and every output line is written contiguously. A type-2 transform grows
// from pass to pass (N -> nf per dimension), so its strided side should be the small input, not the
// large output: with the scatter-out form the last 2-D pass wrote the 33.5 MB fine grid in 64-byte
// pieces (39.6 us at 2048^2 against 21.5 us for the type-1 pass that READS those 33.5 MB).
template <typename T, int LOGN, bool PAD, bool GATHER = false>
__global__ __launch_bounds__(kFftMaxThreads, 4) void fft_rotate_kernel(FftPassArgs<T> a) {
  static_assert(!GATHER || PAD, "the gather form is the type-2 pass");
  using V = typename C2<T>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int n = 1 << LOGN;
  constexpr int TL = n >> 3;              // threads per line
  constexpr int RAD0 = fft_radix<LOGN, 0>();
  constexpr int NP = fft_npass<LOGN>();
  constexpr int RADL = fft_radix<LOGN, NP - 1>();
  const int LW = a.LW;                    // lines in flight
  constexpr int LB = n + (n >> 5) + 1;
  // tile row pitch (+1: the transposed read-out is conflict free). GATHER: a line is transformed IN
  // its tile row (padded line-buffer layout, pitch LB), so no separate line buffers exist and all
  // the LDS goes to rows: 2048-point lines run four at a time (1024 threads) instead of one
  const int TS = GATHER ? LB : a.kout + 1;
  V* lbuf = reinterpret_cast<V*>(smem_raw);               // [LW][LB]
  V* tile = GATHER ? lbuf : lbuf + (size_t)LW * LB;       // [R][TS]
  V* twl = tile + (size_t)a.R * TS;                        // [n / 2]: twiddles (a global table costs an L2 latency per pass)
  const int tid = threadIdx.x;
  for (int i = tid; i < n / 2; i += blockDim.x) twl[i] = a.tw[i];
  const int lw = tid / TL, lt = tid - lw * TL;
  const int64_t line0 = (int64_t)blockIdx.x * a.R;
  const V* in = a.in + (int64_t)blockIdx.y * a.in_batch;
  V* __restrict__ out = a.out + (int64_t)blockIdx.y * a.out_batch;
  const V* twp = twl;
  const T sgn = (T)a.sgn;
  V* mybuf = lbuf + (size_t)lw * LB;   // (GATHER: re-pointed at the line's tile row below)
  const int inlen = PAD ? a.kin : n;

  BinMap<T> map;
  if constexpr (PAD) {          // first-pass element q of this thread <- mode index / factor
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int u = q / RAD0, t = q % RAD0;
      int ak = 0;
      map.idx[q] = bin_to_mode_index(lt + u * TL + t * (n / RAD0), n, a.kin, &ak);
      if (GATHER && map.idx[q] >= 0) map.idx[q] = lpad(map.idx[q]);   // (position in the padded tile row)
      map.sc[q] = a.rf[ak];
    }
  } else {                      // last-pass output q of this thread -> mode index / factor
    constexpr int NsL = n / RADL;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int u = q / RADL, t = q % RADL;
      const int j = lt + u * TL;
      const int k = j & (NsL - 1);
      int ak = 0;
      map.idx[q] = bin_to_mode_index((j - k) * RADL + k + t * NsL, n, a.kout, &ak);
      map.sc[q] = a.rf[ak];
    }
  }

  if constexpr (GATHER) {
    // tile[r][m] = in[m nlines + line0 + r]: consecutive lanes take the R lines of one element
    const int rs = __builtin_ctz(a.R);          // R is a power of two
    const int nthr = blockDim.x;
    for (int e = tid; e < (a.kin << rs); e += nthr) {
      const int m = e >> rs, r = e & (a.R - 1);
      const int64_t line = line0 + r;
      V z; z.x = (T)0; z.y = (T)0;
      if (line < a.nlines) z = in[(int64_t)m * a.nlines + line];
      tile[r * TS + lpad(m)] = z;
    }
  }
  __syncthreads();   // twiddle table (and the gathered lines)
  V v[8], vn[8];
  if constexpr (!GATHER) {
    const int64_t line = line0 + lw;
    const bool live = lw < a.R && line < a.nlines;
    fft_load_line<T, RAD0, PAD>(in + (live ? line : 0) * inlen, live, lt, TL, n, map, v, a.zero_in != 0);
  }
  for (int g = 0; g < a.R; g += LW) {
    const int r = g + lw;                       // line inside the workgroup's block
    const bool more = g + LW < a.R;
    if constexpr (GATHER) {                     // first-pass inputs come from the tile row (modes, padded by the map)
      const bool live = r < a.R && line0 + r < a.nlines;
      fft_load_line<T, RAD0, PAD>(tile + (r < a.R ? r : 0) * TS, live, lt, TL, n, map, v);
    } else if (more) {                          // next group's loads are in flight during this group's passes
      const int64_t line = line0 + r + LW;
      const bool live = r + LW < a.R && line < a.nlines;
      fft_load_line<T, RAD0, PAD>(in + (live ? line : 0) * inlen, live, lt, TL, n, map, vn, a.zero_in != 0);
    }
    V* trow = tile + (r < a.R ? r : 0) * TS;
    if constexpr (GATHER) mybuf = trow;
    // (opaque copy of the lane's index: otherwise the compiler hoists the LDS and twiddle
    // addresses of every pass out of this loop and spills them: 70-280 bytes per lane)
    int ltv = lt;
    asm volatile("" : "+v"(ltv));
    fft_all_passes<T, LOGN, 0, !PAD, GATHER>(twp, map, v, mybuf, trow, ltv, r < a.R, sgn);
    if constexpr (!GATHER) {
      if (more) {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = vn[q];
      }
    }
  }
  __syncthreads();
  const int R = a.R;
  const int64_t nl = a.nlines;
  const int nthreads = blockDim.x;
  if constexpr (GATHER) {
    // ---- tile -> out[line][bin]: whole lines, contiguous (kout = n here)
    for (int e = tid; e < n * R; e += nthreads) {
      const int r = e >> LOGN, bin = e & (n - 1);
      const int64_t line = line0 + r;
      if (line < nl) out[line * n + bin] = tile[r * TS + lpad(bin)];
    }
  } else {
    // ---- tile -> out[bin][line]: R consecutive lines per bin
    for (int e = tid; e < a.kout * R; e += nthreads) {
      const int bin = e / R, r = e - bin * R;
      const int64_t line = line0 + r;
      if (line < nl) out[(int64_t)bin * nl + line] = tile[r * TS + bin];
    }
  }
}

template <typename T, int LOGN>
hipError_t launch_fft_pass(const FftPassArgs<T>& a, bool pad, bool gather, unsigned nblk, unsigned batch, size_t lds,
                           hipStream_t stream) {
  if (pad && a.kout != a.n) return hipErrorInvalidValue;   // a pass either pads or crops
  if (gather != pad) return hipErrorInvalidValue;          // (type 2 = padding passes in gather form, type 1 = cropping scatter-out passes)
  const void* fn = gather ? reinterpret_cast<const void*>(fft_rotate_kernel<T, LOGN, true, true>)
                          : reinterpret_cast<const void*>(fft_rotate_kernel<T, LOGN, false>);
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  if (gather) fft_rotate_kernel<T, LOGN, true, true><<<dim3(nblk, batch), a.LW * (a.n / 8), lds, stream>>>(a);
  else fft_rotate_kernel<T, LOGN, false><<<dim3(nblk, batch), a.LW * (a.n / 8), lds, stream>>>(a);
  return hipGetLastError();
}

// Lines per workgroup (R), lines in flight (LW) and dynamic LDS of one pass; R = 0: not
// supported. Longer output segments (R consecutive lines per bin) write faster; more, smaller
// workgroups fill the chip when a pass has few lines (2048 lines of 2048 points at config 2).
int fft_pass_shape(int n, int kout, int csize, int64_t nlines, int* lw_out, size_t* lds, bool gather = false) {
  if (n < 16 || n > 2048 || (n & (n - 1))) return 0;
  const int TL = n / 8;
  int best = 0, best_lw = 0;
  size_t best_lds = 0;
  int tier = 0;
  int force_r = 0;
#ifdef NUFFT_MIX_SHAPE_ENV   // (experiment builds: lines per workgroup of the power-of-two passes from the environment)
  if (const char* ev = getenv(gather ? "NUFFT_FFT_R_GATHER" : "NUFFT_FFT_R")) force_r = atoi(ev);
#endif
  // Type-2 (gather) passes over very many lines read R lines x 8 bytes per element of a line, the elements nlines x 8
  // bytes apart: from ~2 MB between them every run sits on a page of its own and the longest run that fits wins
  // (r06, profiles/r06_fft_gather_r.txt: 1024^3 fine, R = 8 -> 16: 12.4 -> 9.3 ms; 512^3, R = 8 -> 32: 1.05 -> 0.97 ms)
  const bool far = gather && nlines * (int64_t)csize >= ((int64_t)2 << 20);
  for (int R = 32; R >= 2; R /= 2) {
    if (R > 16 && force_r != R && !far) continue;
    if (force_r && R != force_r) continue;
    if (R * csize < 32) break;                                     // output segments of at least 32 bytes
    int LW = kFftMaxThreads / TL;
    if (LW > R) LW = R;
    if (LW < 1) LW = 1;
    // fewer lines in flight when the tile of R lines leaves no room for more line buffers
    // (2048-point lines with all 2048 bins written: 8 x 16 KB of tile + ONE 16 KB buffer)
    size_t bytes = 0;
    for (; LW >= 1; LW /= 2) {
      bytes = gather ? ((size_t)R * lpad_len(n) + (size_t)n / 2) * csize   // lines are transformed in their tile rows
                     : ((size_t)LW * lpad_len(n) + (size_t)R * (kout + 1) + (size_t)n / 2) * csize;
      if (bytes <= 160 * 1024) break;
      if (gather) { LW = 0; break; }
    }
    if (LW < 1 || LW * TL < 64) continue;                          // at least one full wavefront
    const int64_t wgs = (nlines + R - 1) / R;
    // tier 4: >= 512 workgroups that fit two to a CU, 64-byte segments; tier 3: >= 256 workgroups
    // and 64-byte segments; tier 2: >= 256 workgroups; tier 1: feasible
    const bool seg64 = R * csize >= 64;
    int t = (wgs >= 512 && bytes <= 80 * 1024 && seg64) ? 4 : (wgs >= 256 && seg64) ? 3 : (wgs >= 256 ? 2 : 1);
    if (far && wgs >= 512) t = 5;   // (the first, i.e. longest, feasible R)
    if (t > tier) { tier = t; best = R; best_lw = LW; best_lds = bytes; }
  }
  if (!best) return 0;
  *lw_out = best_lw;
  *lds = best_lds;
  return best;
}


// ------------------------------------------------------------------------------------------
// Mixed-radix passes (r06): every even fine-grid size 2^a 3^b 5^c the reference's
// next_smooth_int picks (nufft_util.cc:119-133; plan: nufft_plan.h:803-863, cuFFT plan
// nufft_plan.cu.cc:2227-2285), not only the powers of two. Same data flow as above -- one
// dimension per launch, output transposed, crop x 1/phi_hat (type 1) / zero-pad (type 2) fused,
// the first type-1 pass zeroing the fine grid -- but the line transform is a Stockham autosort
// over a RUNTIME list of radices out of {2, 3, 4, 5, 6, 8, 10} (6 and 10 as Good-Thomas prime-factor
// butterflies: index permutations known at compile time, no inner twiddles). The R lines of a workgroup
// are transformed IN PLACE in LDS, all at once: a pass's butterflies -- n / radix per line over the R
// lines -- are dealt to the threads as one flat range (every radix fills the threads whatever n / radix
// is); a thread reads its butterflies into registers, the workgroup meets, it writes them back. The
// first pass of a type-1 line reads global memory directly (consecutive lanes, consecutive 8-byte
// elements), the last one writes the cropped, scaled modes as rows [R][kout + 1] into the same LDS,
// which leave transposed; type 2 mirrors it (rows of gathered modes in, the last pass writes whole
// fine-grid lines). R n elements of LDS per workgroup: several workgroups share a CU.
// ------------------------------------------------------------------------------------------
constexpr int kMixMaxPass = 6;
constexpr int kMixMaxN = 4096;

template <typename V, typename T>
__device__ __forceinline__ void dft3(V& v0, V& v1, V& v2, T sgn) {
  const T s = (T)0.86602540378443864676;
  const V t1 = cadd(v1, v2);
  V t2; t2.x = v0.x - (T)0.5 * t1.x; t2.y = v0.y - (T)0.5 * t1.y;
  V d = csub(v1, v2);
  d.x *= s; d.y *= s;
  const V id = mul_i<V, T>(d, sgn);
  v0 = cadd(v0, t1);
  v1 = cadd(t2, id);
  v2 = csub(t2, id);
}
template <typename V, typename T>
__device__ __forceinline__ void dft5(V& v0, V& v1, V& v2, V& v3, V& v4, T sgn) {
  const T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410;
  const T s1 = (T)0.95105651629515357212, s2 = (T)0.58778525229247312917;
  const V a1 = cadd(v1, v4), a2 = cadd(v2, v3), b1 = csub(v1, v4), b2 = csub(v2, v3);
  V r1, r2, i1, i2;
  r1.x = v0.x + c1 * a1.x + c2 * a2.x; r1.y = v0.y + c1 * a1.y + c2 * a2.y;
  r2.x = v0.x + c2 * a1.x + c1 * a2.x; r2.y = v0.y + c2 * a1.y + c1 * a2.y;
  i1.x = s1 * b1.x + s2 * b2.x; i1.y = s1 * b1.y + s2 * b2.y;
  i2.x = s2 * b1.x - s1 * b2.x; i2.y = s2 * b1.y - s1 * b2.y;
  const V j1 = mul_i<V, T>(i1, sgn), j2 = mul_i<V, T>(i2, sgn);
  v0.x += a1.x + a2.x; v0.y += a1.y + a2.y;
  v1 = cadd(r1, j1); v4 = csub(r1, j1);
  v2 = cadd(r2, j2); v3 = csub(r2, j2);
}

// DFT of RAD values in registers, natural order in and out.
template <typename V, typename T, int RAD>
__device__ __forceinline__ void dft_small(V (&v)[RAD], T sgn) {
  if constexpr (RAD == 2) {
    dft2<V, T>(v[0], v[1]);
  } else if constexpr (RAD == 3) {
    dft3<V, T>(v[0], v[1], v[2], sgn);
  } else if constexpr (RAD == 4) {
    dft4<V, T>(v[0], v[1], v[2], v[3], sgn);
  } else if constexpr (RAD == 5) {
    dft5<V, T>(v[0], v[1], v[2], v[3], v[4], sgn);
  } else if constexpr (RAD == 8) {
    dft8<V, T>(v, sgn);
  } else if constexpr (RAD == 6) {
    // Good-Thomas 2 x 3: input (3 n1 + 2 n2) mod 6, output (3 k1 + 4 k2) mod 6
    V a0 = v[0], a1 = v[2], a2 = v[4];   // n1 = 0
    V b0 = v[3], b1 = v[5], b2 = v[1];   // n1 = 1
    dft3<V, T>(a0, a1, a2, sgn);
    dft3<V, T>(b0, b1, b2, sgn);
    v[0] = cadd(a0, b0); v[3] = csub(a0, b0);
    v[4] = cadd(a1, b1); v[1] = csub(a1, b1);
    v[2] = cadd(a2, b2); v[5] = csub(a2, b2);
  } else if constexpr (RAD == 10) {
    // Good-Thomas 2 x 5: input (5 n1 + 2 n2) mod 10, output (5 k1 + 6 k2) mod 10
    V a0 = v[0], a1 = v[2], a2 = v[4], a3 = v[6], a4 = v[8];   // n1 = 0
    V b0 = v[5], b1 = v[7], b2 = v[9], b3 = v[1], b4 = v[3];   // n1 = 1
    dft5<V, T>(a0, a1, a2, a3, a4, sgn);
    dft5<V, T>(b0, b1, b2, b3, b4, sgn);
    v[0] = cadd(a0, b0); v[5] = csub(a0, b0);
    v[6] = cadd(a1, b1); v[1] = csub(a1, b1);
    v[2] = cadd(a2, b2); v[7] = csub(a2, b2);
    v[8] = cadd(a3, b3); v[3] = csub(a3, b3);
    v[4] = cadd(a4, b4); v[9] = csub(a4, b4);
  } else {
    static_assert(RAD == 2, "radix not implemented");
  }
}

// q = j / d for j < 2^16, d < 2^12 with m = floor((2^32 - 1) / d) + 1 (d > 1)
__device__ __forceinline__ unsigned magic_of(unsigned d) { return 0xFFFFFFFFu / d + 1u; }

template <typename T>
struct MixCtx {
  using V = typename C2<T>::type;
  const V* gin;        // type 1: the workgroup's first line (global, n elements each)
  V* gout;             // type 2: the workgroup's first line (global, n elements each)
  V* buf;              // LDS lines [R][LB] (padded layout), transformed in place; the rows of kept modes
                       // ([R][TS]: type-2 input, type-1 output) live in the same memory
  const V* tw;         // LDS, n / 2 twiddles
  const T* rf;         // LDS, reciprocal Fourier series of the kept modes
  int n, K, LB, TS, rows, zero_in;
  T sgn;
};

// One pass of radix RAD over the workgroup's lines, in place: every thread takes its butterflies' inputs into
// registers (at most VPT values rounded up to whole butterflies), the workgroup meets, every thread writes its
// outputs. `first`: inputs from global memory (type 1) or from the rows of modes through the zero-padding map
// (type 2); `last`: outputs to the rows of cropped modes (type 1) or to global memory (type 2).
template <typename T, int RAD, int VPT, bool T2>
__device__ __forceinline__ void mix_pass(const MixCtx<T>& c, int Ns, bool first, bool last) {
  using V = typename C2<T>::type;
  constexpr int ITER = (VPT + RAD - 1) / RAD;
  const int n = c.n;
  const int nb = n / RAD;
  const unsigned m_nb = magic_of((unsigned)nb);
  const int total = c.rows * nb;
  V v[ITER][RAD];
  int jj[ITER], ll[ITER];
  // (opaque copy of the thread index: the passes sit in a loop, and the compiler would otherwise hoist every
  // radix's index arithmetic and addresses out of it and spill them)
  int tidv = threadIdx.x;
  asm volatile("" : "+v"(tidv));
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int b = tidv + it * blockDim.x;
    const int l = nb > 1 ? (int)__umulhi((unsigned)b, m_nb) : b;
    const int j = b - l * nb;
    jj[it] = j;
    ll[it] = b < total ? l : -1;
    if (b >= total) continue;
    if (first) {
      if constexpr (T2) {
        const V* row = c.buf + l * c.TS;
#pragma unroll
        for (int t = 0; t < RAD; ++t) {
          int ak = 0;
          const int idx = bin_to_mode_index(j + t * nb, n, c.K, &ak);
          V y; y.x = (T)0; y.y = (T)0;
          if (idx >= 0) {
            const V z = row[idx];
            const T sc = c.rf[ak];
            y.x = z.x * sc; y.y = z.y * sc;
          }
          v[it][t] = y;
        }
      } else {
        const V* row = c.gin + (int64_t)l * n;
#pragma unroll
        for (int t = 0; t < RAD; ++t) v[it][t] = row[j + t * nb];
        if (c.zero_in) {
          V zero; zero.x = (T)0; zero.y = (T)0;
          V* w = const_cast<V*>(row);
#pragma unroll
          for (int t = 0; t < RAD; ++t) w[j + t * nb] = zero;
        }
      }
    } else {
      const V* row = c.buf + l * c.LB;
#pragma unroll
      for (int t = 0; t < RAD; ++t) v[it][t] = row[lpad(j + t * nb)];
    }
  }
  if (!first || T2) __syncthreads();   // every input of the pass is in registers
  const unsigned m_ns = Ns > 1 ? magic_of((unsigned)Ns) : 0u;
  const int tstep = nb / Ns;   // n / (Ns RAD)
  const int half = n >> 1;
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int l = ll[it], j = jj[it];
    if (l < 0) continue;
    int k = 0;
    if (Ns > 1) {
      k = j - (int)__umulhi((unsigned)j, m_ns) * Ns;
      const int base = k * tstep;
#pragma unroll
      for (int t = 1; t < RAD; ++t) {
        int ti = t * base;                 // < n
        const bool neg = ti >= half;
        if (neg) ti -= half;
        V w = c.tw[ti];
        if (neg) { w.x = -w.x; w.y = -w.y; }
        v[it][t] = cmul<V, T>(v[it][t], w);
      }
    }
    dft_small<V, T, RAD>(v[it], c.sgn);
    const int o = (j - k) * RAD + k;
    if (last) {
      if constexpr (T2) {
        V* row = c.gout + (int64_t)l * n;
#pragma unroll
        for (int t = 0; t < RAD; ++t) row[o + t * Ns] = v[it][t];
      } else {
        V* row = c.buf + l * c.TS;
#pragma unroll
        for (int t = 0; t < RAD; ++t) {
          int ak = 0;
          const int idx = bin_to_mode_index(o + t * Ns, n, c.K, &ak);
          if (idx >= 0) {
            const T sc = c.rf[ak];
            V y; y.x = v[it][t].x * sc; y.y = v[it][t].y * sc;
            row[idx] = y;
          }
        }
      }
    } else {
      V* row = c.buf + l * c.LB;
#pragma unroll
      for (int t = 0; t < RAD; ++t) row[lpad(o + t * Ns)] = v[it][t];
    }
  }
  if (!(last && T2)) __syncthreads();
}

// T2 = false: type-1 pass (lines of n points in, kout modes out, transposed). T2 = true: type-2 pass
// (kin modes in, interleaved [element][line]; whole lines of n points out). blockDim.x >= R n / VPT.
template <typename T, int VPT, bool T2>
__global__ __launch_bounds__(kFftMaxThreads) void fft_mixed_kernel(FftPassArgs<T> a) {
  using V = typename C2<T>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int n = a.n;
  const int K = T2 ? a.kin : a.kout;        // kept modes of this dimension
  const int LB = lpad_len(n);
  const int TS = K + 1;
  const int R = a.R;
  V* buf = reinterpret_cast<V*>(smem_raw);                  // [R][LB]
  V* twl = buf + (size_t)R * LB;                            // [n / 2]
  T* rfl = reinterpret_cast<T*>(twl + (n >> 1));            // [K / 2 + 1]
  const int tid = threadIdx.x, nthr = blockDim.x;
  for (int i = tid; i < (n >> 1); i += nthr) twl[i] = a.tw[i];
  for (int i = tid; i <= (K >> 1); i += nthr) rfl[i] = a.rf[i];
  const int64_t line0 = (int64_t)blockIdx.x * R;
  const V* in = a.in + (int64_t)blockIdx.y * a.in_batch;
  V* __restrict__ out = a.out + (int64_t)blockIdx.y * a.out_batch;
  const int64_t nl = a.nlines;
  const int rows = (int)(nl - line0 < R ? nl - line0 : R);   // live lines of this workgroup
  const int rs = __builtin_ctz(R);                            // R is a power of two
  if constexpr (T2) {
    // rows of modes: buf[r][m] = in[m nlines + line0 + r]: consecutive lanes take the R lines of one element
    for (int e = tid; e < (K << rs); e += nthr) {
      const int m = e >> rs, r = e & (R - 1);
      if (r < rows) buf[r * TS + m] = in[(int64_t)m * nl + line0 + r];
    }
    __syncthreads();
  }
  MixCtx<T> c;
  c.buf = buf; c.tw = twl; c.rf = rfl;
  c.n = n; c.K = K; c.LB = LB; c.TS = TS; c.zero_in = a.zero_in; c.rows = rows;
  c.sgn = (T)a.sgn;
  c.gin = T2 ? nullptr : in + line0 * n;
  c.gout = T2 ? out + line0 * n : nullptr;
  int Ns = 1;
  for (int p = 0; p < a.npass; ++p) {
    const int rad = (int)((a.radpack >> (5 * p)) & 31u);
    const bool first = p == 0, last = p == a.npass - 1;
    switch (rad) {
      case 2: mix_pass<T, 2, VPT, T2>(c, Ns, first, last); break;
      case 3: mix_pass<T, 3, VPT, T2>(c, Ns, first, last); break;
      case 4: mix_pass<T, 4, VPT, T2>(c, Ns, first, last); break;
      case 5: mix_pass<T, 5, VPT, T2>(c, Ns, first, last); break;
      case 6: mix_pass<T, 6, VPT, T2>(c, Ns, first, last); break;
      case 8: mix_pass<T, 8, VPT, T2>(c, Ns, first, last); break;
      case 10: mix_pass<T, 10, VPT, T2>(c, Ns, first, last); break;
      default: break;
    }
    Ns *= rad;
  }
  if constexpr (!T2) {
    // ---- rows of modes -> out[bin][line]: R consecutive lines per bin
    for (int e = tid; e < (K << rs); e += nthr) {
      const int bin = e >> rs, r = e & (R - 1);
      if (r < rows) out[(int64_t)bin * nl + line0 + r] = buf[r * TS + bin];
    }
  }
}

// Radices of the passes of an n-point line, fewest passes first, then the largest smallest radix;
// largest radix first (the first pass multiplies no twiddles). 0: n is not 2^a 3^b 5^c.
int mix_factor(int n, unsigned* radpack) {
  static const int kRad[] = {10, 8, 6, 5, 4, 3, 2};
  int best[kMixMaxPass], bestn = 0, cur[kMixMaxPass];
  // depth-first over non-increasing radix lists
  struct Rec {
    static void go(int rem, int depth, int maxi, int* cur, int* best, int* bestn) {
      if (rem == 1) {
        bool better = *bestn == 0 || depth < *bestn;
        if (!better && depth == *bestn) {
          // same number of passes: prefer the larger smallest radix, then the larger largest
          if (cur[depth - 1] != best[depth - 1]) better = cur[depth - 1] > best[depth - 1];
          else better = cur[0] > best[0];
        }
        if (better) { *bestn = depth; for (int i = 0; i < depth; ++i) best[i] = cur[i]; }
        return;
      }
      if (depth >= kMixMaxPass || (*bestn && depth >= *bestn)) return;
      for (int i = maxi; i < 7; ++i) {
        if (rem % kRad[i]) continue;
        cur[depth] = kRad[i];
        go(rem / kRad[i], depth + 1, i, cur, best, bestn);
      }
    }
  };
  Rec::go(n, 0, 0, cur, best, &bestn);
  if (!bestn) return 0;
  unsigned pk = 0;
  for (int i = 0; i < bestn; ++i) pk |= (unsigned)best[i] << (5 * i);
  *radpack = pk;
  return bestn;
}

// Lines per workgroup (R, a power of two), threads, values per thread (8 / 16) and dynamic LDS of one mixed-radix
// pass with K kept modes; 0: not supported. The R lines are all in flight: R n elements of LDS and R n / 8 (or / 16)
// threads, so that several workgroups share a CU (the loads of one overlap the passes of another).
int mix_pass_shape(int n, int K, int csize, int64_t nlines, int* threads, int* vpt_out, size_t* lds, bool gather = false) {
  unsigned pk;
  if (n < 4 || n > kMixMaxN || (n & 1) || K > n || !mix_factor(n, &pk)) return 0;
  int best = 0, best_thr = 0, best_vpt = 0;
  double tier = 0.0;
  size_t best_lds = 0;
  int force_r = 0;
#ifdef NUFFT_MIX_SHAPE_ENV   // (experiment builds, tools/exp_mixfft_shape.py: lines per workgroup from the environment)
  if (const char* ev = getenv("NUFFT_MIX_R")) force_r = atoi(ev);
#endif
  for (int R = 16; R >= 1; R /= 2) {
    if (force_r && R != force_r) continue;
    if (R * csize < 32 && best) break;                               // output segments of at least 32 bytes where anything else fits
    const size_t bytes = ((size_t)R * lpad_len(n) + (size_t)n / 2) * csize + ((size_t)K / 2 + 1) * (csize / 2);
    if (bytes > 160 * 1024) continue;
    int vpt = 8;
    int thr = (int)(((int64_t)R * n + 8 * 64 - 1) / (8 * 64)) * 64;
    if (thr > kFftMaxThreads) { vpt = 16; thr = (int)(((int64_t)R * n + 16 * 64 - 1) / (16 * 64)) * 64; }
    if (thr > kFftMaxThreads) continue;
    const int64_t wgs = (nlines + R - 1) / R;
    const int per_cu = std::min((int)(160 * 1024 / bytes), 1536 / thr);   // (72-80 VGPRs: six waves per SIMD)
    const bool seg64 = R * csize >= 64;
    // (the far-apart-gather rule of fft_pass_shape does not carry over: one workgroup of R = 16 lines per CU instead of
    // three of 8 lost more than the longer runs gained -- 640^3 fine type 2 2.65 -> 2.96 ms, 768^3 3.7 -> 5.2 ms; `gather` unused)
    (void)gather;
    // (fewer workgroups than CUs: the more the better -- the second pass of a 1920^2 grid has 960 lines, and eight of them per
    // workgroup left half the chip idle: 22 us for 22 MB, profiles/r06_configs_kernel_stats.txt)
    const double t = (wgs >= 512 && per_cu >= 2 && seg64) ? 4.0 : (wgs >= 256 && seg64) ? 3.0 : (wgs >= 256 ? 2.0 : 1.0 + 0.9 * (double)wgs / 256.0);
    if (t > tier) { tier = t; best = R; best_thr = thr; best_vpt = vpt; best_lds = bytes; }
  }
  if (!best) return 0;
  *threads = best_thr;
  *vpt_out = best_vpt;
  *lds = best_lds;
  return best;
}

template <typename T, int VPT>
hipError_t launch_mixed_pass_v(const FftPassArgs<T>& a, bool t2, unsigned nblk, unsigned batch, int threads, size_t lds,
                               hipStream_t stream) {
  const void* fn = t2 ? reinterpret_cast<const void*>(fft_mixed_kernel<T, VPT, true>)
                      : reinterpret_cast<const void*>(fft_mixed_kernel<T, VPT, false>);
  if (lds > 64 * 1024) {
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  if (t2) fft_mixed_kernel<T, VPT, true><<<dim3(nblk, batch), threads, lds, stream>>>(a);
  else fft_mixed_kernel<T, VPT, false><<<dim3(nblk, batch), threads, lds, stream>>>(a);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_mixed_pass(FftPassArgs<T> a, bool t2, unsigned nblk, unsigned batch, int threads, int vpt, size_t lds,
                             hipStream_t stream) {
  a.npass = mix_factor(a.n, &a.radpack);
  if (!a.npass) return hipErrorInvalidValue;
  return vpt == 8 ? launch_mixed_pass_v<T, 8>(a, t2, nblk, batch, threads, lds, stream)
                  : launch_mixed_pass_v<T, 16>(a, t2, nblk, batch, threads, lds, stream);
}

inline bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

}  // namespace

bool pruned_fft_supported(const Geom& g, int precision) {
  if (g.tuning & NUFFT_HIP_TUNE_ROCFFT) return false;   // second opinion / A/B: rocFFT + deconvolve kernel
  const int csize = 2 * precision;
  for (int d = 0; d < g.rank; ++d) {
    size_t lds;
    int lw;
    if (g.nmodes[d] > g.nf[d]) return false;
    if (!is_pow2(g.nf[d]) || g.nf[d] > 2048) {
      // r06: mixed-radix passes for the other smooth sizes
      int vpt;
      if ((g.tuning & NUFFT_HIP_TUNE_MIXFFT_OFF) || !mix_pass_shape(g.nf[d], g.nmodes[d], csize, 1 << 20, &lw, &vpt, &lds)) return false;
      continue;
    }
    // type 1 crops to nmodes, type 2 writes all nf bins: both shapes must fit
    if (!fft_pass_shape(g.nf[d], g.nmodes[d], csize, 1 << 20, &lw, &lds)) return false;
    if (!fft_pass_shape(g.nf[d], g.nf[d], csize, 1 << 20, &lw, &lds)) return false;
    if (!fft_pass_shape(g.nf[d], g.nf[d], csize, 1 << 20, &lw, &lds, true)) return false;
  }
  return true;
}

// Complex elements (per transform) each of the two intermediate buffers must hold.
int64_t pruned_fft_tmp_elems(const Geom& g) {
  if (g.rank < 2) return 0;
  int64_t best = 0;
  // type 1: B1 = N0 nf1 nf2, B2 = N1 N0 nf2; type 2: B1 = nf0 N1 N2, B2 = nf1 nf0 N2
  const int64_t N0 = g.nmodes[0], N1 = g.nmodes[1], N2 = g.nmodes[2];
  const int64_t f0 = g.nf[0], f1 = g.nf[1], f2 = g.nf[2];
  best = std::max(best, N0 * f1 * f2);
  best = std::max(best, f0 * N1 * N2);
  if (g.rank > 2) {
    best = std::max(best, N1 * N0 * f2);
    best = std::max(best, f1 * f0 * N2);
  }
  return best;
}

template <typename T>
hipError_t launch_pruned_fft(const Geom& g, int type, int iflag, T* fine, T* f, T* tmp0, T* tmp1,
                             const T* const rf[3], const T* const tw[3], int batch, hipStream_t stream,
                             bool zero_fine) {
  using V = typename C2<T>::type;
  const int rank = g.rank;
  const int csize = 2 * (int)sizeof(T);
  int64_t fine_elems = 1, mode_elems = 1;
  for (int d = 0; d < rank; ++d) { fine_elems *= g.nf[d]; mode_elems *= g.nmodes[d]; }
  // sizes of the array before pass d: dims [0..d) already transformed (type 1: cropped to
  // nmodes, type 2: expanded to nf), dims [d..rank) not yet
  const V* src = reinterpret_cast<const V*>(type == 1 ? fine : f);
  int64_t src_batch = type == 1 ? fine_elems : mode_elems;
  // Type 1 walks the dimensions fastest first: contiguous input lines, output transposed [bin][line]
  // (the next dimension's lines are then contiguous) -- the strided side is the shrinking output.
  // Type 2 (gather form) walks them SLOWEST first: the dimension's lines lie interleaved in the
  // input ([element][line]: the strided, still small side) and leave contiguous ([line][bin]);
  // after rank passes the layout is back to x fastest either way.
  const bool gather = type == 2;   // (r02: 2048^2 type 2 59 -> 43 us against the scatter-out form)
  for (int step = 0; step < rank; ++step) {
    const int d = gather ? rank - 1 - step : step;
    FftPassArgs<T> a;
    a.n = g.nf[d];
    a.kin = type == 1 ? g.nf[d] : g.nmodes[d];
    a.kout = type == 1 ? g.nmodes[d] : g.nf[d];
    int64_t lines = 1;
    for (int e = 0; e < rank; ++e) {
      if (e == d) continue;
      const bool done = gather ? e > d : e < d;
      lines *= (type == 1) ? (done ? g.nmodes[e] : g.nf[e]) : (done ? g.nf[e] : g.nmodes[e]);
    }
    a.nlines = lines;
    a.in = src;
    a.in_batch = src_batch;
    const bool last = step == rank - 1;
    V* dst = last ? reinterpret_cast<V*>(type == 1 ? f : fine) : reinterpret_cast<V*>((step & 1) ? tmp1 : tmp0);
    a.out = dst;
    a.out_batch = (int64_t)a.kout * lines;
    a.tw = reinterpret_cast<const V*>(tw[d]);
    a.rf = rf[d];
    a.sgn = iflag < 0 ? -1.0f : 1.0f;
    a.zero_in = (zero_fine && type == 1 && step == 0) ? 1 : 0;
    a.npass = 0;
    a.radpack = 0;
    size_t lds = 0;
    a.LW = 1;
    const bool mixed = !is_pow2(a.n) || a.n > 2048;
    int mix_threads = 0, mix_vpt = 0;
    a.R = mixed ? mix_pass_shape(a.n, type == 1 ? a.kout : a.kin, csize, lines, &mix_threads, &mix_vpt, &lds, gather)
                : fft_pass_shape(a.n, a.kout, csize, lines, &a.LW, &lds, gather);
    if (a.R == 0) return hipErrorInvalidValue;
    const int64_t nblk = (lines + a.R - 1) / a.R;
    if (nblk > 2147483647LL || batch > 65535) return hipErrorInvalidValue;
    hipError_t e = hipErrorInvalidValue;
    if (mixed) {
      e = launch_mixed_pass<T>(a, type == 2, (unsigned)nblk, (unsigned)batch, mix_threads, mix_vpt, lds, stream);
      if (e != hipSuccess) return e;
      src = dst;
      src_batch = a.out_batch;
      continue;
    }
    switch (a.n) {
      case 16: e = launch_fft_pass<T, 4>(a, type == 2, gather, (unsigned)nblk, (unsigned)batch, lds, stream); break;
      case 32: e = launch_fft_pass<T, 5>(a, type == 2, gather, (unsigned)nblk, (unsigned)batch, lds, stream); break;
      case 64: e = launch_fft_pass<T, 6>(a, type == 2, gather, (unsigned)nblk, (unsigned)batch, lds, stream); break;
      case 128: e = launch_fft_pass<T, 7>(a, type == 2, gather, (unsigned)nblk, (unsigned)batch, lds, stream); break;
      case 256: e = launch_fft_pass<T, 8>(a, type == 2, gather, (unsigned)nblk, (unsigned)batch, lds, stream); break;
      case 512: e = launch_fft_pass<T, 9>(a, type == 2, gather, (unsigned)nblk, (unsigned)batch, lds, stream); break;
      case 1024: e = launch_fft_pass<T, 10>(a, type == 2, gather, (unsigned)nblk, (unsigned)batch, lds, stream); break;
      case 2048: e = launch_fft_pass<T, 11>(a, type == 2, gather, (unsigned)nblk, (unsigned)batch, lds, stream); break;
      default: break;
    }
    if (e != hipSuccess) return e;
    src = dst;
    src_batch = a.out_batch;
  }
  return hipGetLastError();
}
template hipError_t launch_pruned_fft<float>(const Geom&, int, int, float*, float*, float*, float*,
                                             const float* const[3], const float* const[3], int, hipStream_t, bool);
template hipError_t launch_pruned_fft<double>(const Geom&, int, int, double*, double*, double*, double*,
                                              const double* const[3], const double* const[3], int, hipStream_t, bool);

}  // namespace nufft_hip
