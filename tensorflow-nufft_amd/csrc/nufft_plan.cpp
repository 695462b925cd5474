// Host side of the gfx950 NUFFT plan: parameter rules, kernel tables, device
// workspace, rocFFT plan, and the stage sequencing of set_points / execute /
// spread / interp behind the C ABI of include/nufft_hip.h.
//
// Replaces Plan<GPUDevice, FloatType> of the reference
// (tensorflow_nufft/cc/kernels/nufft_plan.cu.cc: initialize :1808-2030,
// set_points :2054-2111, execute :2113-2168, interp/spread :2170-2225,
// setup_spreader :3040-3099, set_grid_size :3166-3204) together with the
// shared rules in nufft_plan.h:739-863 and nufft_util.cc:43-117.
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "nufft_hip_internal.h"

using namespace nufft_hip;

namespace {

constexpr double kPi = 3.14159265358979323846;
constexpr int64_t kMaxArraySize = 2000000000;  // reference nufft_plan.h:62

std::string format(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return std::string(buf);
}

// ------------------------------------------------------------- numerics

// Smallest even integer >= n with no prime factor above 5
// (reference next_smooth_integer, nufft_plan.h:628-649).
int64_t next_smooth_even(int64_t n) {
  if (n <= 2) return 2;
  if (n & 1) ++n;
  for (;; n += 2) {
    int64_t d = n;
    while (d % 2 == 0) d /= 2;
    while (d % 3 == 0) d /= 3;
    while (d % 5 == 0) d /= 5;
    if (d == 1) return n;
  }
}

struct KernelSpec {
  int w = 0;
  double beta = 0, c = 0, sigma = 2.0;
  // Normalised exponential-of-semicircle kernel, phi(0) = 1:
  //   phi(x) = exp(beta (sqrt(1 - c x^2) - 1)),  |x| <= w/2.
  // The reference leaves it unnormalised (exp(beta) ~ 1e8 at w = 8,
  // nufft_util.cc:64-69); the constant cancels in the deconvolution, and the
  // normalised form cannot overflow float in 3D at large w.
  double eval(double x) const {
    const double t = 1.0 - c * x * x;
    return t <= 0.0 ? (std::fabs(x) <= 0.5 * w ? std::exp(-beta) : 0.0)
                    : std::exp(beta * (std::sqrt(t) - 1.0));
  }
};

// Gauss-Legendre rule by Newton iteration on the Legendre recurrence (own code;
// the reference uses the LGPL legendre_compute_glr, legendre_rule_fast.cc:28).
void gauss_legendre(int n, std::vector<double>& z, std::vector<double>& wt) {
  z.resize(n);
  wt.resize(n);
  for (int i = 0; i < n; ++i) {
    double x = std::cos(kPi * (i + 0.75) / (n + 0.5));
    double dp = 1.0;
    for (int it = 0; it < 100; ++it) {
      double p0 = 1.0, p1 = x;
      for (int k = 2; k <= n; ++k) {
        const double pk = ((2.0 * k - 1.0) * x * p1 - (k - 1.0) * p0) / k;
        p0 = p1;
        p1 = pk;
      }
      dp = n * (x * p1 - p0) / (x * x - 1.0);
      const double dx = p1 / dp;
      x -= dx;
      if (std::fabs(dx) < 1e-16) break;
    }
    double p0 = 1.0, p1 = x;
    for (int k = 2; k <= n; ++k) {
      const double pk = ((2.0 * k - 1.0) * x * p1 - (k - 1.0) * p0) / k;
      p0 = p1;
      p1 = pk;
    }
    dp = n * (x * p1 - p0) / (x * x - 1.0);
    z[i] = x;
    wt[i] = 2.0 / ((1.0 - x * x) * dp * dp);
  }
}

// Fourier series of the kernel on an nf-point grid, k = 0..nf/2, including the
// (-1)^k that cancels the +pi shift of the fold (reference kernel_fseries_1d,
// nufft_util.cc:71-117; q = floor(2 + 3 w / 2) positive nodes). Always double.
void kernel_fseries(const KernelSpec& ks, int64_t nf, std::vector<double>& out) {
  const double hw = 0.5 * ks.w;
  const int q = (int)(2 + 3.0 * hw);
  std::vector<double> z, wt;
  gauss_legendre(2 * q, z, wt);
  out.assign(nf / 2 + 1, 0.0);
  for (int64_t k = 0; k <= nf / 2; ++k) {
    double s = 0.0;
    for (int n = 0; n < q; ++n) {
      const double zn = z[n] * hw;
      s += 2.0 * hw * wt[n] * ks.eval(zn) *
           std::cos(2.0 * kPi * (double)k * ((double)(nf / 2) - zn) / (double)nf);
    }
    out[k] = s;
  }
}

// Spread-/interp-only normalisation (reference calculate_scale_factor,
// nufft_util.cc:43-62), for the NORMALISED kernel: the reference multiplies by
// 1 / (h (1 + sum exp(beta sqrt(1 - t_i^2))) w/2)^rank with its unnormalised
// kernel, so ours carries an extra exp(beta)^rank.
double spread_only_scale(const KernelSpec& ks, int rank) {
  const int n = 100;
  const double h = 2.0 / n;
  double x = -1.0, sum = 0.0;
  for (int i = 1; i < n; ++i) {
    x += h;
    sum += std::exp(ks.beta * (std::sqrt(1.0 - x * x) - 1.0));
  }
  sum += std::exp(-ks.beta);
  sum *= h;
  sum *= std::sqrt(1.0 / ks.c);
  double scale = sum;
  if (rank > 1) scale *= sum;
  if (rank > 2) scale *= sum;
  return 1.0 / scale;
}

// Piecewise polynomial (one per stencil cell j) in z in [-1, 1]:
// P_j(z) ~ phi((z + 1 - w)/2 + j). Chebyshev interpolation -> monomial
// coefficients, all in double. Plays the role of the reference's generated
// kernel_horner_*.inc tables (not copied; any sigma works). Returns the max
// error over a dense sample; *max_abs = the largest |P_j(z)| on that sample (the
// fit overshoots phi(0) = 1 by about its own error, e.g. 1.00001 at w = 4).
double fit_horner(const KernelSpec& ks, int nc, double* tab /* [kMaxCoef][kMaxW] */, double* max_abs = nullptr) {
  const int w = ks.w;
  std::vector<double> node(nc), val(nc), cheb(nc);
  for (int i = 0; i < kMaxCoef * kMaxW; ++i) tab[i] = 0.0;
  for (int j = 0; j < w; ++j) {
    for (int i = 0; i < nc; ++i) {
      node[i] = std::cos(kPi * (i + 0.5) / nc);
      val[i] = ks.eval((node[i] + 1.0 - w) / 2.0 + j);
    }
    for (int k = 0; k < nc; ++k) {
      double s = 0;
      for (int i = 0; i < nc; ++i) s += val[i] * std::cos(kPi * k * (i + 0.5) / nc);
      cheb[k] = s * (k == 0 ? 1.0 : 2.0) / nc;
    }
    std::vector<double> mono(nc, 0.0), tkm1(nc, 0.0), tk(nc, 0.0), tn(nc, 0.0);
    tkm1[0] = 1.0;
    if (nc > 1) tk[1] = 1.0;
    for (int m = 0; m < nc; ++m) mono[m] += cheb[0] * tkm1[m];
    if (nc > 1)
      for (int m = 0; m < nc; ++m) mono[m] += cheb[1] * tk[m];
    for (int k = 2; k < nc; ++k) {
      for (int m = 0; m < nc; ++m) tn[m] = (m > 0 ? 2.0 * tk[m - 1] : 0.0) - tkm1[m];
      for (int m = 0; m < nc; ++m) {
        mono[m] += cheb[k] * tn[m];
        tkm1[m] = tk[m];
        tk[m] = tn[m];
      }
    }
    for (int k = 0; k < nc; ++k) tab[k * kMaxW + j] = mono[k];
  }
  double err = 0.0, kmax = 0.0;
  for (int s = 0; s <= 64; ++s) {
    const double z = -1.0 + 2.0 * s / 64.0;
    for (int j = 0; j < w; ++j) {
      double acc = tab[(nc - 1) * kMaxW + j];
      for (int k = nc - 2; k >= 0; --k) acc = acc * z + tab[k * kMaxW + j];
      err = std::max(err, std::fabs(acc - ks.eval((z + 1.0 - w) / 2.0 + j)));
      kmax = std::max(kmax, std::fabs(acc));
    }
  }
  if (max_abs) *max_abs = kmax;
  return err;
}

std::once_flag g_rocfft_once;

// tune_mode() reads the bits from a Geom: a Geom that holds nothing but them
Geom g_tuning_probe(int tuning) {
  Geom t{};
  t.tuning = tuning;
  return t;
}

}  // namespace

// ------------------------------------------------------------------ plan

struct nufft_hip_plan_s {
  int type = 0, rank = 0, iflag = 0, ntransf = 0, precision = 0;
  double tol = 0;
  nufft_hip_options opts;
  Geom g;
  KernelSpec ks;
  int method = NUFFT_HIP_METHOD_TILE_GENERIC;
  int batch_size = 1;
  size_t lds_bytes = 0;
  hipStream_t stream = nullptr;
  int device = 0;
  double spread_scale = 1.0;

  std::vector<double> horner_h;      // [kMaxCoef][kMaxW]
  std::vector<double> fser_h[3];     // phi-hat per dimension
  void* d_horner = nullptr;
  void* d_rfser[3] = {nullptr, nullptr, nullptr};
  void* d_fine = nullptr;
  int64_t fine_elems = 0, grid_elems = 0;
  std::map<int, rocfft_plan> fft_plans;   // by batch count (batch_size and the remainder batch)
  rocfft_execution_info fft_info = nullptr;
  void* fft_work = nullptr;
  size_t fft_work_bytes = 0;      // bytes the FFT plans need (max over them)
  bool own_fft = false;           // pruned per-dimension passes (nufft_fft.hip) instead of rocFFT + deconvolve
  int64_t fine_clear_slots = 0;   // leading fine grids known to be all zero (left so by the last type-1 FFT pass)
  void* d_twiddle[3] = {nullptr, nullptr, nullptr};   // exp(iflag 2 pi i m / nf_d)
  void* fft_tmp[2] = {nullptr, nullptr};              // intermediates of the pruned passes

  int nitems = 1;                // point sets handled together (options.num_point_sets)
  int64_t M = 0, cap = 0, cap_global = 0, cap_tile_of = 0;   // M: points per set
  void* rec = nullptr;           // Rec<T>[cap], tile-sorted
  void* rec2 = nullptr;          // Rec<T>[cap2]: target of the lazy cell-sort pass (then swapped with rec)
  int64_t cap2 = 0;
  int64_t spread_uses = 0;       // spread launches fed by the current sorted points
  int32_t *hist = nullptr;       // LDS-histogram sort: [nblk][ntiles]
  int64_t hist_elems = 0;
  int32_t *tile_of = nullptr, *rank_of = nullptr;   // global-counter sort
  int32_t *tile_count = nullptr, *tile_start = nullptr, *sub_start = nullptr, *bad_count = nullptr;
  float* cstats = nullptr;       // 3-D float fixed-point plans: {largest, summed} strength of every slot of a spread launch
  int64_t cap_cstats = 0;        //   (+ the per-workgroup partials of the reduction: sized per point count)
  float* sub_bound = nullptr;    // Geom::fx_patch: count-filter bound per subproblem
  int64_t cap_sub_bound = 0;
  int* fb_list = nullptr;        // fixed-point 3-D plans: count + launch slots of the subproblems left to the fp64 planes
  int64_t cap_fb_list = 0;
  const PointsIn* direct_in = nullptr;   // r06: set for the duration of a direct (unsorted) type-2 call of the one-call entry
  int4* segs = nullptr;          // Geom::stack: [0].x = how many stacks, [1..] their descriptors (stack_plan_kernel)
  int64_t cap_segs = 0;
  TapMax taps = {};              // per-tap maxima of the fitted kernel (bound3_kernel)
  int64_t workspace_bytes = 0;
  bool points_set = false;
  bool host_only = false;        // nufft_hip_plan_create_host: no device state at all
  bool fused = false;            // current records carry the strengths (execute_with_points)
  int stop_after = -1;           // debug: execute returns after this stage
  nufft_hip_allocator allocator = {nullptr, nullptr, nullptr};   // workspace allocator (null: hipMalloc)
  bool fixed_ws = false;         // tile tables / fine grid / FFT work buffer are allocated
  std::string err;

  // optional per-stage timing with HIP events on the plan's stream
  int timing = 0;   // 0 off, 1 every stage, 2 only the spread / interp kernel
  struct Pending { int stage; hipEvent_t e0, e1; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> free_events;
  hipEvent_t open_event[STAGE_COUNT] = {};
  double stage_ms[STAGE_COUNT] = {};
  int stage_calls[STAGE_COUNT] = {};
};

namespace {

#define HIP_TRY(p, expr)                                                              \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess) {                                                           \
      (p)->err = format("HIP error %d (%s) at %s:%d", (int)e_, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                          \
      return NUFFT_HIP_INTERNAL;                                                      \
    }                                                                                 \
  } while (0)

#define FFT_TRY(p, expr)                                                      \
  do {                                                                        \
    rocfft_status s_ = (expr);                                                \
    if (s_ != rocfft_status_success) {                                        \
      (p)->err = format("rocFFT error %d at %s:%d", (int)s_, __FILE__, __LINE__); \
      return NUFFT_HIP_INTERNAL;                                              \
    }                                                                         \
  } while (0)

// Debug allocator ("electric fence", NUFFT_HIP_DEBUG_EFENCE=1): every plan buffer gets its
// own mapping inside a larger reserved address range and ENDS within 15 bytes of the end of
// that mapping, so the first access past its end (and any access before its first page)
// faults instead of silently touching a neighbour. GPU AddressSanitizer is not available
// on this stack; tests/test_gpu_parity.py::test_plan_buffers_under_electric_fence runs a
// mix of plans this way. Freed ranges stay reserved (see fence_free): with recycled addresses
// the r02 investigation saw corrupted results and spurious faults in long test processes that
// no plan kernel was involved in.
struct FenceRec { void* va; size_t reserved; void* mapped_at; size_t mapped; hipMemGenericAllocationHandle_t handle; };
static std::map<void*, FenceRec> g_fence;
static std::mutex g_fence_mu;
static bool fence_enabled() {
  static const bool on = getenv("NUFFT_HIP_DEBUG_EFENCE") != nullptr;
  return on;
}
static hipError_t fence_alloc(void** ptr, size_t bytes) {
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  prop.location.id = dev;
  size_t gran = 0;
  e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum);
  if (e != hipSuccess) return e;
  if (!gran) return hipErrorInvalidValue;
  FenceRec r = {};
  r.mapped = (bytes + gran - 1) / gran * gran;
  r.reserved = r.mapped + 2 * gran;            // one unmapped granule on either side
  if ((e = hipMemAddressReserve(&r.va, r.reserved, gran, nullptr, 0)) != hipSuccess) return e;
  if ((e = hipMemCreate(&r.handle, r.mapped, &prop, 0)) != hipSuccess) return e;
  r.mapped_at = (char*)r.va + gran;
  if ((e = hipMemMap(r.mapped_at, r.mapped, 0, r.handle, 0)) != hipSuccess) return e;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  if ((e = hipMemSetAccess(r.mapped_at, r.mapped, &acc, 1)) != hipSuccess) return e;
  *ptr = (char*)r.mapped_at + ((r.mapped - bytes) & ~(size_t)15);
  std::lock_guard<std::mutex> lk(g_fence_mu);
  g_fence[*ptr] = r;
  return hipSuccess;
}
static bool fence_free(void* p) {
  FenceRec r;
  {
    std::lock_guard<std::mutex> lk(g_fence_mu);
    auto it = g_fence.find(p);
    if (it == g_fence.end()) return false;
    r = it->second;
    g_fence.erase(it);
  }
  (void)hipDeviceSynchronize();
  (void)hipMemUnmap(r.mapped_at, r.mapped);
  (void)hipMemRelease(r.handle);
  // The address range is NOT given back: a later mapping at a recycled address was seen to go wrong
  // intermittently (results of plans created right after differently-sized ones were freed; never with
  // hipMalloc), and a range that stays reserved and unmapped also turns any use after free into a fault.
  return true;
}

// Workspace buffers (fine grid, sorted records, sort tables, FFT work buffer) come from
// the plan's allocator when the host framework supplied one (include/nufft_hip.h,
// nufft_hip_allocator), else from hipMalloc. The constant tables use table_alloc.
int dev_alloc(nufft_hip_plan p, void** ptr, size_t bytes) {
  *ptr = nullptr;
  if (bytes == 0) bytes = 16;
  if (p->allocator.alloc) {
    *ptr = p->allocator.alloc(bytes, p->allocator.user);
    if (!*ptr) {
      p->err = format("out of device memory allocating %zu bytes (host allocator)", bytes);
      return NUFFT_HIP_RESOURCE_EXHAUSTED;
    }
    p->workspace_bytes += (int64_t)bytes;
    return NUFFT_HIP_OK;
  }
  hipError_t e = fence_enabled() ? fence_alloc(ptr, bytes) : hipMalloc(ptr, bytes);
  if (e != hipSuccess) {
    p->err = format("out of device memory allocating %zu bytes (%s)", bytes, hipGetErrorString(e));
    (void)hipGetLastError();
    return NUFFT_HIP_RESOURCE_EXHAUSTED;
  }
  p->workspace_bytes += (int64_t)bytes;
  return NUFFT_HIP_OK;
}

void dev_free(nufft_hip_plan p, void* ptr) {
  if (!ptr) return;
  if (p->allocator.alloc) {
    if (p->allocator.free) p->allocator.free(ptr, p->allocator.user);
    return;
  }
  if (fence_enabled() && fence_free(ptr)) return;
  (void)hipFree(ptr);
}

int table_alloc(nufft_hip_plan p, void** ptr, size_t bytes) {
  *ptr = nullptr;
  const hipError_t e = hipMalloc(ptr, bytes ? bytes : 16);
  if (e != hipSuccess) {
    p->err = format("out of device memory allocating %zu bytes (%s)", bytes, hipGetErrorString(e));
    (void)hipGetLastError();
    return NUFFT_HIP_RESOURCE_EXHAUSTED;
  }
  return NUFFT_HIP_OK;
}

// Growing a buffer: with hipMalloc the old block may still be in use by queued kernels, so
// the stream is drained first (hipFree would synchronise the device anyway); a framework
// allocator orders reuse behind the stream itself.
int sync_before_regrow(nufft_hip_plan p) {
  if (p->allocator.alloc) return NUFFT_HIP_OK;
  HIP_TRY(p, hipStreamSynchronize(p->stream));
  return NUFFT_HIP_OK;
}

// Both FFT plans a plan can need (full batches and the remainder batch) are built at plan
// creation, so that execute never creates a plan, allocates or synchronises.
int build_fft_plan(nufft_hip_plan p, int batch) {
  if (batch <= 0 || p->fft_plans.count(batch)) return NUFFT_HIP_OK;
  std::call_once(g_rocfft_once, [] { rocfft_setup(); });
  size_t lengths[3];
  for (int d = 0; d < p->rank; ++d) lengths[d] = (size_t)p->g.nf[d];
  rocfft_plan plan = nullptr;
  FFT_TRY(p, rocfft_plan_create(
                 &plan, rocfft_placement_inplace,
                 p->iflag < 0 ? rocfft_transform_type_complex_forward
                              : rocfft_transform_type_complex_inverse,
                 p->precision == NUFFT_HIP_F32 ? rocfft_precision_single : rocfft_precision_double,
                 (size_t)p->rank, lengths, (size_t)batch, nullptr));
  p->fft_plans[batch] = plan;
  size_t wb = 0;
  FFT_TRY(p, rocfft_plan_get_work_buffer_size(plan, &wb));
  p->fft_work_bytes = std::max(p->fft_work_bytes, wb);
  if (!p->fft_info) FFT_TRY(p, rocfft_execution_info_create(&p->fft_info));
  FFT_TRY(p, rocfft_execution_info_set_stream(p->fft_info, p->stream));
  return NUFFT_HIP_OK;
}

// Fixed-size part of the workspace: tile tables, the fine grid and the FFT work buffer.
// Allocated at plan creation (internal allocator) or at the first set_points after a
// release (framework allocator).
int ensure_fixed_workspace(nufft_hip_plan p) {
  if (p->fixed_ws) return NUFFT_HIP_OK;
  const Geom& g = p->g;
  int rc = dev_alloc(p, (void**)&p->tile_count, sizeof(int32_t) * (size_t)g.ntiles);
  if (!rc) rc = dev_alloc(p, (void**)&p->tile_start, sizeof(int32_t) * ((size_t)g.ntiles + 2));   // (+1: most subproblems of a tile)
  if (!rc) rc = dev_alloc(p, (void**)&p->sub_start, sizeof(int32_t) * ((size_t)g.ntiles + 1));
  if (!rc) rc = dev_alloc(p, (void**)&p->bad_count, sizeof(int32_t) * 4);
  if (!rc && !p->opts.spread_only)
    rc = dev_alloc(p, &p->d_fine, (size_t)p->precision * 2 * (size_t)p->fine_elems * p->batch_size * p->nitems);
  if (!rc && p->own_fft) {
    const size_t tmp_bytes = (size_t)p->precision * 2 * (size_t)pruned_fft_tmp_elems(p->g) * p->batch_size * p->nitems;
    if (p->rank >= 2) rc = dev_alloc(p, &p->fft_tmp[0], tmp_bytes);
    if (!rc && p->rank >= 3) rc = dev_alloc(p, &p->fft_tmp[1], tmp_bytes);
  }
  if (!rc && !p->opts.spread_only && p->fft_work_bytes) {
    rc = dev_alloc(p, &p->fft_work, p->fft_work_bytes);
    if (!rc) FFT_TRY(p, rocfft_execution_info_set_work_buffer(p->fft_info, p->fft_work, p->fft_work_bytes));
  }
  if (rc) return rc;
  p->fixed_ws = true;
  return NUFFT_HIP_OK;
}

void release_workspace(nufft_hip_plan p) {
  void** bufs[] = {(void**)&p->tile_count, (void**)&p->tile_start, (void**)&p->sub_start, (void**)&p->bad_count,
                   &p->d_fine, &p->fft_work, &p->fft_tmp[0], &p->fft_tmp[1], &p->rec, &p->rec2, (void**)&p->hist, (void**)&p->tile_of,
                   (void**)&p->rank_of, (void**)&p->cstats, (void**)&p->sub_bound, (void**)&p->fb_list, (void**)&p->segs};
  for (void** b : bufs) {
    dev_free(p, *b);
    *b = nullptr;
  }
  p->cap = p->cap2 = p->cap_global = p->cap_tile_of = p->cap_sub_bound = p->cap_cstats = p->cap_fb_list = p->cap_segs = 0;
  p->hist_elems = 0;
  p->workspace_bytes = 0;
  p->fixed_ws = false;
  p->fine_clear_slots = 0;
  p->points_set = false;
  p->fused = false;
  p->M = 0;
}

template <typename T>
int upload_tables(nufft_hip_plan p) {
  std::vector<T> tmp(kMaxCoef * kMaxW);
  for (size_t i = 0; i < tmp.size(); ++i) tmp[i] = (T)p->horner_h[i];
  int rc = table_alloc(p, &p->d_horner, tmp.size() * sizeof(T));
  if (rc) return rc;
  HIP_TRY(p, hipMemcpy(p->d_horner, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice));
  for (int d = 0; d < p->rank; ++d) {
    const size_t n = p->fser_h[d].size();
    std::vector<T> r(n);
    for (size_t k = 0; k < n; ++k) r[k] = (T)(1.0 / p->fser_h[d][k]);
    rc = table_alloc(p, &p->d_rfser[d], n * sizeof(T));
    if (rc) return rc;
    HIP_TRY(p, hipMemcpy(p->d_rfser[d], r.data(), n * sizeof(T), hipMemcpyHostToDevice));
  }
  if (p->own_fft) {
    for (int d = 0; d < p->rank; ++d) {
      const int64_t n = p->g.nf[d];
      std::vector<T> tw(2 * (size_t)n);
      for (int64_t m = 0; m < n; ++m) {
        const double ang = (double)p->iflag * 2.0 * kPi * (double)m / (double)n;
        tw[2 * m] = (T)std::cos(ang);
        tw[2 * m + 1] = (T)std::sin(ang);
      }
      rc = table_alloc(p, &p->d_twiddle[d], tw.size() * sizeof(T));
      if (rc) return rc;
      HIP_TRY(p, hipMemcpy(p->d_twiddle[d], tw.data(), tw.size() * sizeof(T), hipMemcpyHostToDevice));
    }
  }
  return NUFFT_HIP_OK;
}

hipEvent_t take_event(nufft_hip_plan p) {
  if (!p->free_events.empty()) {
    hipEvent_t e = p->free_events.back();
    p->free_events.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
void stage_begin(void* ctx, int stage) {
  nufft_hip_plan p = (nufft_hip_plan)ctx;
  if (p->timing == 2 && stage != STAGE_SPREAD && stage != STAGE_INTERP) return;
  hipEvent_t e = take_event(p);
  (void)hipEventRecord(e, p->stream);
  p->open_event[stage] = e;
}
void stage_end(void* ctx, int stage) {
  nufft_hip_plan p = (nufft_hip_plan)ctx;
  if (p->timing == 2 && stage != STAGE_SPREAD && stage != STAGE_INTERP) return;
  hipEvent_t e = take_event(p);
  (void)hipEventRecord(e, p->stream);
  p->pending.push_back({stage, p->open_event[stage], e});
}
StageHook make_hook(nufft_hip_plan p) {
  StageHook h;
  if (p->timing) {
    h.ctx = p;
    h.begin_fn = stage_begin;
    h.end_fn = stage_end;
  }
  return h;
}

// Most (transform, point set) slots one spread launch of this plan covers: execute passes batch_size transforms,
// the spread-only entry up to 32768 (spread_interp_impl)
int spread_slots(nufft_hip_plan p) {
  return (p->opts.spread_only ? std::min(32768, p->ntransf) : p->batch_size) * std::max(1, p->nitems);
}

// rec_mult: record slots per point (2 for the 32-byte fused records of 3-D float plans)
int ensure_point_capacity(nufft_hip_plan p, int64_t M, int rec_mult = 1) {
  int rc = ensure_fixed_workspace(p);
  if (rc) return rc;
  const int mode = sort_mode(p->g, M);
  const int items = p->g.nitems > 1 ? p->g.nitems : 1;
  int64_t need = 0, per_block;
  if (mode == 0) need = (int64_t)sort_blocks(p->g, M, &per_block) * items * p->g.ntiles;
  if (mode == 1) {
    const int64_t nblk = (int64_t)sort_blocks16(p->g, M, &per_block) * items;
    need = nblk * ((p->g.ntiles + 1) / 2) + nblk * (int64_t)p->g.ntiles;   // hist16 words + 32-bit prefix
  }
  if (mode == 3) need = sort2_layout(p->g, M).words;
  if (need > p->hist_elems) {
    if ((rc = sync_before_regrow(p))) return rc;
    dev_free(p, p->hist);
    p->hist = nullptr;
    p->hist_elems = 0;
    if ((rc = dev_alloc(p, (void**)&p->hist, sizeof(int32_t) * (size_t)need))) return rc;
    p->hist_elems = need;
  }
  if ((mode == 1 || mode == 2) && M > p->cap_global) {   // per-point rank array (16-bit ranks in mode 1)
    if ((rc = sync_before_regrow(p))) return rc;
    dev_free(p, p->rank_of);
    p->rank_of = nullptr;
    p->cap_global = 0;
    if ((rc = dev_alloc(p, (void**)&p->rank_of, (size_t)M * 4))) return rc;
    p->cap_global = M;
  }
  if (mode == 2 && M > p->cap_tile_of) {   // per-point tile array: the global-counter path only
    if ((rc = sync_before_regrow(p))) return rc;
    dev_free(p, p->tile_of);
    p->tile_of = nullptr;
    p->cap_tile_of = 0;
    if ((rc = dev_alloc(p, (void**)&p->tile_of, (size_t)M * 4))) return rc;
    p->cap_tile_of = M;
  }
  if (p->g.fixed_point && p->rank == 3 && (p->type == NUFFT_HIP_TYPE_1 || p->opts.spread_only)) {
    const int64_t need_s = (int64_t)cstats_floats(M / std::max(1, p->nitems), spread_slots(p));
    if (need_s > p->cap_cstats) {
      if ((rc = sync_before_regrow(p))) return rc;
      dev_free(p, p->cstats);
      p->cstats = nullptr;
      p->cap_cstats = 0;
      if ((rc = dev_alloc(p, (void**)&p->cstats, sizeof(float) * (size_t)need_s))) return rc;
      p->cap_cstats = need_s;
    }
  }
  if (p->g.fixed_point && p->rank == 3 && (p->type == NUFFT_HIP_TYPE_1 || p->opts.spread_only)) {
    const int64_t need_l = (int64_t)subproblem_grid_bound(p->g, M) + 1;
    if (need_l > p->cap_fb_list) {
      if ((rc = sync_before_regrow(p))) return rc;
      dev_free(p, p->fb_list);
      p->fb_list = nullptr;
      p->cap_fb_list = 0;
      if ((rc = dev_alloc(p, (void**)&p->fb_list, sizeof(int) * (size_t)need_l))) return rc;
      p->cap_fb_list = need_l;
    }
  }
  // fixed-point 3-D plans that spread over stacks of tiles (r05): the stack descriptors
  // (type-2 plans: the double-precision w <= 8 interpolation walks stacks too, r06)
  const bool t2_stacks = p->g.fp64_stack && (p->g.wide || p->precision == NUFFT_HIP_F64);   // (and the w = 9..16 interpolation; the float
                                                                                              // interpolation over stacks lost at every density, r05 and r06)
  const bool stacks = (p->g.fixed_point || p->g.fp64_stack) && p->rank == 3 && (p->type == NUFFT_HIP_TYPE_1 || p->opts.spread_only || t2_stacks) && stack3_wanted(p->g, M);
  if (stacks) {
    const int64_t need_g = (int64_t)stack_grid_bound(p->g, M) + 1;
    if (need_g > p->cap_segs) {
      if ((rc = sync_before_regrow(p))) return rc;
      dev_free(p, p->segs);
      p->segs = nullptr;
      p->cap_segs = 0;
      if ((rc = dev_alloc(p, (void**)&p->segs, sizeof(int4) * (size_t)need_g))) return rc;
      p->cap_segs = need_g;
    }
  }
  if (p->g.fx_patch && (p->type == NUFFT_HIP_TYPE_1 || p->opts.spread_only)) {
    // (a plan that spreads over stacks keeps one bound per stack in the same buffer)
    const int64_t need_b = (int64_t)std::max(subproblem_grid_bound(p->g, M), stacks ? stack_grid_bound(p->g, M) : 0u) + 1;
    if (need_b > p->cap_sub_bound) {
      if ((rc = sync_before_regrow(p))) return rc;
      dev_free(p, p->sub_bound);
      p->sub_bound = nullptr;
      p->cap_sub_bound = 0;
      if ((rc = dev_alloc(p, (void**)&p->sub_bound, sizeof(float) * (size_t)need_b))) return rc;
      p->cap_sub_bound = need_b;
    }
  }
  const int64_t slots = M * rec_mult;
  if (slots <= p->cap) return NUFFT_HIP_OK;
  if ((rc = sync_before_regrow(p))) return rc;
  dev_free(p, p->rec);
  p->rec = nullptr;
  p->cap = 0;
  const size_t rec_bytes = p->precision == NUFFT_HIP_F32 ? sizeof(Rec<float>) : sizeof(Rec<double>);
  if ((rc = dev_alloc(p, &p->rec, (size_t)slots * rec_bytes))) return rc;
  p->cap = slots;
  return NUFFT_HIP_OK;
}

// `strengths` non-null: fused sort (the records carry them; the caller has checked
// fused_sort_supported).
template <typename T>
int set_points_impl(nufft_hip_plan p, int64_t M, const void* x, const void* y, const void* z,
                    int64_t stride, const void* strengths = nullptr) {
  const int64_t Mtot = M * p->nitems;   // M points in each of the nitems sets
  int rc = ensure_point_capacity(p, Mtot, (strengths && p->rank == 3) ? (int)(sizeof(FusedRec3) / sizeof(Rec<float>)) : 1);
  if (rc) return rc;
  p->M = M;
  p->points_set = false;
  const bool check = p->opts.check_points_range && p->opts.points_range != NUFFT_HIP_RANGE_INFINITE;
  if (check) HIP_TRY(p, hipMemsetAsync(p->bad_count, 0, sizeof(int32_t) * 4, p->stream));
  PointsIn in;
  // unused dimensions alias x so that the kernels can load all three unconditionally
  in.pts[0] = x; in.pts[1] = p->rank > 1 ? y : x; in.pts[2] = p->rank > 2 ? z : x;
  in.stride = stride;
  in.M = Mtot;
  in.M_item = M;
  in.blocks_per_item = 1;
  in.range_mode = p->opts.points_range;
  in.check_range = check ? 1 : 0;
  in.strengths = strengths;
  // one interleaved [M, rank] array with x last (the op's layout): vector loads in the sort
  in.aos = 0;
  if (p->rank >= 2 && stride == p->rank && M >= 2) {
    const char* base = (const char*)(p->rank == 2 ? y : z);
    const bool cols = (const char*)x == base + (size_t)(p->rank - 1) * p->precision &&
                      (p->rank == 2 || (const char*)y == base + p->precision);
    const size_t align = p->rank == 2 ? 4 * (size_t)p->precision : (size_t)p->precision;   // point PAIRS in 2-D
    // (paired float loads start every set on a pair: an odd set size keeps them for one set only)
    const bool pair_ok = !(p->rank == 2 && p->precision == NUFFT_HIP_F32 && p->nitems > 1 && (M & 1));
    if (cols && pair_ok && ((uintptr_t)base % align) == 0) in.aos = p->rank;
  }
  SortWork w;
  w.hist = p->hist; w.tile_of = p->tile_of; w.rank_of = p->rank_of;
  w.tile_count = p->tile_count; w.tile_start = p->tile_start; w.sub_start = p->sub_start;
  w.bad_count = p->bad_count;
  p->g.cell_sorted = 0;   // the 2-D spread's second sort level runs lazily (maybe_cellsort)
  p->g.fused = strengths ? 1 : 0;
  p->fused = strengths != nullptr;
  p->spread_uses = 0;
  // 3-D interp plans order every subproblem by start cell right away (rec2 -> rec)
  // (never behind a fused sort: its 32-byte records are consumed by the one type-1 spread they were sorted for,
  // and the cell sort reads and writes 16-byte records)
  const bool cells = !strengths && (p->type == NUFFT_HIP_TYPE_2 || p->opts.spread_only) &&
                     cellsort_wanted_interp(p->g, p->method, p->precision, M);
  if (cells && Mtot > p->cap2) {
    if ((rc = sync_before_regrow(p))) return rc;
    dev_free(p, p->rec2);
    p->rec2 = nullptr;
    p->cap2 = 0;
    if ((rc = dev_alloc(p, &p->rec2, (size_t)Mtot * sizeof(Rec<T>)))) return rc;
    p->cap2 = Mtot;
  }
  SortedOut<T> out;
  out.rec = (Rec<T>*)(cells ? p->rec2 : p->rec);
  w.tmp = nullptr;
  if (sort_mode(p->g, Mtot) == 3) {
    // level-1 records of the two-level sort: in the buffer the sort does NOT end in -- the cell-sort target
    // where there is one (it is written last), else rec2
    if (cells) {
      w.tmp = p->rec;
    } else {
      const int64_t slots2 = Mtot;
      if (slots2 > p->cap2) {
        if ((rc = sync_before_regrow(p))) return rc;
        dev_free(p, p->rec2);
        p->rec2 = nullptr;
        p->cap2 = 0;
        if ((rc = dev_alloc(p, &p->rec2, (size_t)slots2 * sizeof(Rec<T>)))) return rc;
        p->cap2 = slots2;
      }
      w.tmp = p->rec2;
    }
  }
  const StageHook hook = make_hook(p);
  HIP_TRY(p, launch_sort<T>(p->g, in, w, out, p->stream, hook));
  if (cells) {
    hook.begin(STAGE_SORT_CELL);
    HIP_TRY(p, launch_cellsort<T>(p->g, Mtot, p->tile_start, p->sub_start, (const Rec<T>*)p->rec2,
                                  (Rec<T>*)p->rec, p->stream));
    hook.end(STAGE_SORT_CELL);
  }
  // stacks of tiles (r05): 3-D fixed-point float plans spread over them
  p->g.stack = (sizeof(T) == 4 && p->g.fixed_point && p->rank == 3 && (!p->g.fx_patch || p->sub_bound) && p->segs && Mtot > 0 &&
                (p->type == NUFFT_HIP_TYPE_1 || p->opts.spread_only) && stack3_wanted(p->g, Mtot) &&
                p->cap_segs >= (int64_t)stack_grid_bound(p->g, Mtot) + 1 &&   // (r05 advisor: the descriptor buffer is this large, not merely present)
                (!p->g.fx_patch || p->cap_sub_bound >= (int64_t)stack_grid_bound(p->g, Mtot) + 1)) ? 1 : 0;
  if (sizeof(T) == 8 || p->g.wide) {
    // r06: the fp64-plane kernels over the same stacks (double: spread_wave3_stack_kernel; w = 9..16: spread_wide_kernel)
    p->g.stack = (p->g.fp64_stack && p->segs && Mtot > 0 &&   // (type 1, type 2 and the spread / interp ops alike: every kernel family here walks stacks)
                  p->cap_segs >= (int64_t)stack_grid_bound(p->g, Mtot) + 1 && stack3_wanted(p->g, Mtot)) ? 1 : 0;
    if (p->g.stack) {
      hook.begin(STAGE_SORT_CELL);
      HIP_TRY(p, launch_stack_plan(p->g, p->tile_start, Mtot, p->segs + 1, (int*)p->segs, p->stream));
      hook.end(STAGE_SORT_CELL);
    }
  }
  if constexpr (sizeof(T) == 4) {
    // w = 7, 8 fixed-point plans: the bound that fixes every subproblem's step (and which of them keep fp64 planes)
    if (p->g.stack && !p->g.fx_patch) {
      // w <= 6 (spread_dense3_stack_kernel): the stacks, and the subproblems of crowded tiles for the fp64 planes as before
      hook.begin(STAGE_SORT_CELL);
      HIP_TRY(p, launch_stack_plan(p->g, p->tile_start, Mtot, p->segs + 1, (int*)p->segs, p->stream));
      hook.end(STAGE_SORT_CELL);
      if (p->fb_list) {
        if (M > (int64_t)p->g.fx_max_subs * p->g.max_sub) HIP_TRY(p, launch_crowded_list(p->g, p->sub_start, p->fb_list, p->stream));
        else HIP_TRY(p, hipMemsetAsync(p->fb_list, 0, sizeof(int), p->stream));
      }
    } else if (p->g.stack) {
      // stacks of tiles (spread_stack3_kernel): cut them, then the bound that fixes every stack's step
      hook.begin(STAGE_SORT_CELL);
      HIP_TRY(p, launch_stack_plan(p->g, p->tile_start, Mtot, p->segs + 1, (int*)p->segs, p->stream));
      HIP_TRY(p, launch_bound3_stack(p->g, (const Rec<float>*)p->rec, (int)sizeof(Rec<float>), p->tile_start, p->sub_start, Mtot,
                                     p->taps, p->segs + 1, (const int*)p->segs, p->sub_bound, p->fb_list, p->stream));
      hook.end(STAGE_SORT_CELL);
    } else if (p->g.fx_patch && p->sub_bound && Mtot > 0) {
      hook.begin(STAGE_SORT_CELL);
      HIP_TRY(p, launch_bound3(p->g, (const Rec<float>*)p->rec, (int)sizeof(Rec<float>), p->tile_start, p->sub_start,
                               subproblem_grid_bound(p->g, Mtot), p->taps, p->sub_bound, p->fb_list, p->stream));
      hook.end(STAGE_SORT_CELL);
    } else if (p->g.fixed_point && p->rank == 3 && p->fb_list) {
      // the other fixed-point plans: the subproblems of crowded tiles (none can exist below fx_max_subs full subproblems)
      if (M > (int64_t)p->g.fx_max_subs * p->g.max_sub) HIP_TRY(p, launch_crowded_list(p->g, p->sub_start, p->fb_list, p->stream));
      else HIP_TRY(p, hipMemsetAsync(p->fb_list, 0, sizeof(int), p->stream));
    }
  }
  if (check) {
    int32_t bad = 0;
    HIP_TRY(p, hipMemcpyAsync(&bad, p->bad_count, sizeof(int32_t), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(p, hipStreamSynchronize(p->stream));
    if (bad > 0) {
      const double b = p->opts.points_range == NUFFT_HIP_RANGE_STRICT ? kPi : 3.0 * kPi;
      // message of reference check_points_within_range, nufft_plan.h:889-893
      p->err = format(
          "Found points outside expected range. Valid range is [%g, %g]. Check your points "
          "and/or set a less restrictive value for options.points_range.", -b, b);
      return NUFFT_HIP_INVALID_ARGUMENT;
    }
  }
  p->points_set = true;
  return NUFFT_HIP_OK;
}

template <typename T>
SortedPoints<T> sorted_view(nufft_hip_plan p) {
  SortedPoints<T> sp;
  sp.rec = (const Rec<T>*)p->rec;
  sp.tile_start = p->tile_start;
  sp.sub_start = p->sub_start;
  sp.cstats = p->cstats;
  sp.cstats_slots = spread_slots(p);
  sp.cstats_blocks = p->cstats ? cstats_blocks(p->M, sp.cstats_slots) : 0;
  sp.sub_bound = p->sub_bound;
  sp.fb_list = p->fb_list;
  sp.fb_ticket = p->bad_count ? p->bad_count + 2 : nullptr;   // (bad_count[0]: the range check's flag; [2]: this counter)
  sp.segs = p->g.stack ? p->segs + 1 : nullptr;
  sp.seg_count = (const int*)p->segs;
  sp.seg_bound = p->sub_bound;
  return sp;
}

// type 1: c -> fine grid (spread) -> FFT -> deconvolve -> f
// type 2: f -> amplify into fine grid -> FFT -> interpolate -> c
// batched over min(ntransf, batch_size) transforms per pass (one launch per
// stage and pass; the reference launches per transform, nufft_plan.cu.cc:2469-2571).

// Second sort level, applied lazily. Ordering every subproblem by stencil start
// cell costs about a third of a spread launch and saves about a sixth of each one
// (the cell-grouped kernel then skips its in-LDS sort), so it only pays when the
// same points feed several launches: a plan reused across executes, or many
// transforms per execute. It therefore runs inside the execute that brings the
// count to three, never in set_points, and never while the stream is being
// captured into a graph (it may allocate).
template <typename T>
int maybe_cellsort(nufft_hip_plan p, int launches) {
  const int64_t before = p->spread_uses;
  p->spread_uses += launches;
  if (p->g.cell_sorted || p->fused || before + launches < 3) return NUFFT_HIP_OK;
  if (!cellsort_wanted(p->g, p->method, p->precision, p->M)) return NUFFT_HIP_OK;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(p->stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return NUFFT_HIP_OK;
  const int64_t Mtot = p->M * p->nitems;
  if (Mtot > p->cap2) {
    int rc;
    if ((rc = sync_before_regrow(p))) return rc;
    dev_free(p, p->rec2);
    p->rec2 = nullptr;
    p->cap2 = 0;
    if ((rc = dev_alloc(p, &p->rec2, (size_t)Mtot * sizeof(Rec<T>)))) return rc;
    p->cap2 = Mtot;
  }
  const StageHook hook = make_hook(p);
  hook.begin(STAGE_SORT_CELL);
  HIP_TRY(p, launch_cellsort<T>(p->g, Mtot, p->tile_start, p->sub_start, (const Rec<T>*)p->rec,
                                (Rec<T>*)p->rec2, p->stream));
  hook.end(STAGE_SORT_CELL);
  std::swap(p->rec, p->rec2);
  std::swap(p->cap, p->cap2);
  p->g.cell_sorted = 1;
  return NUFFT_HIP_OK;
}

template <typename T>
int execute_impl(nufft_hip_plan p, void* c, void* f) {
  if (!p->points_set && !p->direct_in) {
    p->err = "set_points must be called before execute";
    return NUFFT_HIP_INVALID_ARGUMENT;
  }
  if (p->type == NUFFT_HIP_TYPE_1) {
    const int rc = maybe_cellsort<T>(p, p->ntransf);
    if (rc) return rc;
  }
  const SortedPoints<T> sp = sorted_view<T>(p);
  const T* rf[3] = {(const T*)p->d_rfser[0], (const T*)p->d_rfser[1], (const T*)p->d_rfser[2]};
  const int stop = p->stop_after;
  for (int b0 = 0; b0 < p->ntransf; b0 += p->batch_size) {
    const int nb = std::min(p->batch_size, p->ntransf - b0);
    T* cb = (T*)c + 2 * (int64_t)b0 * p->M;
    T* fb = (T*)f + 2 * (int64_t)b0 * p->grid_elems;
    T* fw = (T*)p->d_fine;
    const int slots = nb * p->nitems;        // fine grids in flight: every point set x nb transforms
    // pruned passes with the deconvolution fused in, unless a test wants to look at the fine
    // grid between FFT and deconvolution (then the rocFFT path, planned by debug_stop_after)
    const bool own = p->own_fft && stop != STAGE_FFT && stop != STAGE_DECONVOLVE;
    rocfft_plan fft = nullptr;
    if (!own) {
      const auto it = p->fft_plans.find(slots);   // both batch counts were planned at creation
      if (it == p->fft_plans.end()) {
        p->err = format("no FFT plan for a batch of %d transforms", nb);
        return NUFFT_HIP_INTERNAL;
      }
      fft = it->second;
    }
    const T* tw[3] = {(const T*)p->d_twiddle[0], (const T*)p->d_twiddle[1], (const T*)p->d_twiddle[2]};
    void* bufs[1] = {fw};
    const StageHook hook = make_hook(p);
    if (p->type == NUFFT_HIP_TYPE_1) {
      // the pruned FFT's first pass reads the whole fine grid and leaves it zeroed for the next
      // spread: the memset only runs when that is not known to have happened
      if (p->fine_clear_slots < slots) {
        hook.begin(STAGE_ZERO);
        HIP_TRY(p, hipMemsetAsync(fw, 0, sizeof(T) * 2 * (size_t)p->fine_elems * slots, p->stream));
        hook.end(STAGE_ZERO);
      }
      p->fine_clear_slots = 0;   // (dirty from here on, until the FFT pass has cleared it again)
      hook.begin(STAGE_SPREAD);
      HIP_TRY(p, launch_spread<T>(p->g, p->method, sp, p->M * p->nitems, (const T*)p->d_horner, cb, fw, nb,
                                  p->M, p->fine_elems, (T)1, p->lds_bytes, p->stream));
      hook.end(STAGE_SPREAD);
      if (stop == STAGE_SPREAD) continue;
      if (own) {
        hook.begin(STAGE_FFT);
        HIP_TRY(p, launch_pruned_fft<T>(p->g, 1, p->iflag, fw, fb, (T*)p->fft_tmp[0], (T*)p->fft_tmp[1], rf, tw,
                                        slots, p->stream, /*zero_fine=*/true));
        hook.end(STAGE_FFT);
        p->fine_clear_slots = slots;
        continue;
      }
      hook.begin(STAGE_FFT);
      FFT_TRY(p, rocfft_execute(fft, bufs, nullptr, p->fft_info));
      hook.end(STAGE_FFT);
      if (stop == STAGE_FFT) continue;
      hook.begin(STAGE_DECONVOLVE);
      HIP_TRY(p, launch_deconvolve<T>(p->g, 1, fb, fw, rf, slots, p->stream));
      hook.end(STAGE_DECONVOLVE);
    } else {
      if (own) {
        hook.begin(STAGE_FFT);
        HIP_TRY(p, launch_pruned_fft<T>(p->g, 2, p->iflag, fw, fb, (T*)p->fft_tmp[0], (T*)p->fft_tmp[1], rf, tw,
                                        slots, p->stream));
        hook.end(STAGE_FFT);
      } else {
        hook.begin(STAGE_DECONVOLVE);
        HIP_TRY(p, launch_deconvolve<T>(p->g, 2, fb, fw, rf, slots, p->stream));
        hook.end(STAGE_DECONVOLVE);
        if (stop == STAGE_DECONVOLVE) continue;
        hook.begin(STAGE_FFT);
        FFT_TRY(p, rocfft_execute(fft, bufs, nullptr, p->fft_info));
        hook.end(STAGE_FFT);
        if (stop == STAGE_FFT) continue;
      }
      hook.begin(STAGE_INTERP);
      if (p->direct_in)   // r06: straight from the caller's points (small calls of the one-call entry)
        HIP_TRY(p, launch_interp_direct<T>(p->g, *p->direct_in, (const T*)p->d_horner, cb, fw, nb, p->M, p->fine_elems, (T)1, p->stream));
      else
        HIP_TRY(p, launch_interp<T>(p->g, p->method, sp, p->M * p->nitems, (const T*)p->d_horner, cb, fw, nb,
                                    p->M, p->fine_elems, (T)1, p->stream));
      hook.end(STAGE_INTERP);
    }
  }
  return NUFFT_HIP_OK;
}

template <typename T>
int spread_interp_impl(nufft_hip_plan p, int dir, void* c, void* f) {
  if (!p->opts.spread_only) {
    p->err = "spread/interp require a plan created with options.spread_only = 1";
    return NUFFT_HIP_INVALID_ARGUMENT;
  }
  if (!p->points_set) {
    p->err = "set_points must be called before spread/interp";
    return NUFFT_HIP_INVALID_ARGUMENT;
  }
  if (dir == 1) {
    const int rc = maybe_cellsort<T>(p, p->ntransf);
    if (rc) return rc;
  }
  const SortedPoints<T> sp = sorted_view<T>(p);
  const T scale = (T)p->spread_scale;
  // The op output IS the grid here; batch over all transforms in one launch
  // group of at most 65535 (grid.y limit).
  for (int b0 = 0; b0 < p->ntransf; b0 += 32768) {   // (one pass when nitems > 1: ntransf <= 32768 then)
    const int nb = std::min(32768, p->ntransf - b0);
    const int slots = nb * p->nitems;
    T* cb = (T*)c + 2 * (int64_t)b0 * p->M;
    T* fb = (T*)f + 2 * (int64_t)b0 * p->fine_elems;
    const StageHook hook = make_hook(p);
    if (dir == 1) {
      hook.begin(STAGE_ZERO);
      HIP_TRY(p, hipMemsetAsync(fb, 0, sizeof(T) * 2 * (size_t)p->fine_elems * slots, p->stream));
      hook.end(STAGE_ZERO);
      hook.begin(STAGE_SPREAD);
      HIP_TRY(p, launch_spread<T>(p->g, p->method, sp, p->M * p->nitems, (const T*)p->d_horner, cb, fb, nb,
                                  p->M, p->fine_elems, scale, p->lds_bytes, p->stream));
      hook.end(STAGE_SPREAD);
    } else {
      hook.begin(STAGE_INTERP);
      HIP_TRY(p, launch_interp<T>(p->g, p->method, sp, p->M * p->nitems, (const T*)p->d_horner, cb, fb, nb,
                                  p->M, p->fine_elems, scale, p->stream));
      hook.end(STAGE_INTERP);
    }
  }
  return NUFFT_HIP_OK;
}

void destroy(nufft_hip_plan p) {
  if (!p) return;
  if (!p->host_only) {
    // a framework allocator orders reuse behind the stream itself; ours (hipFree) does not
    if (!p->allocator.alloc) (void)hipStreamSynchronize(p->stream);
    release_workspace(p);
    if (!p->fft_plans.empty() || p->d_horner) (void)hipStreamSynchronize(p->stream);   // tables / twiddles are hipFree'd
    for (auto& kv : p->fft_plans) rocfft_plan_destroy(kv.second);
    if (p->fft_info) rocfft_execution_info_destroy(p->fft_info);
    if (p->d_horner) (void)hipFree(p->d_horner);
    for (int d = 0; d < 3; ++d) {
      if (p->d_rfser[d]) (void)hipFree(p->d_rfser[d]);
      if (p->d_twiddle[d]) (void)hipFree(p->d_twiddle[d]);
    }
    for (auto& pe : p->pending) { (void)hipEventDestroy(pe.e0); (void)hipEventDestroy(pe.e1); }
    for (auto e : p->free_events) (void)hipEventDestroy(e);
  }
  delete p;
}

}  // namespace

// ------------------------------------------------------------------ C ABI

extern "C" {

int nufft_hip_abi_version(void) { return NUFFT_HIP_ABI_VERSION; }

void nufft_hip_default_options(nufft_hip_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->points_range = NUFFT_HIP_RANGE_EXTENDED;   // reference default, nufft_options.py
}

// Host-only part of plan creation: argument checks and every parameter rule.
// On success returns a plan object with no device state.
static int configure(nufft_hip_plan* out, int type, int rank, const int64_t* grid_dims,
                     int iflag, int ntransf, double tol, int precision,
                     const nufft_hip_options* opts_in, void* stream, std::string* errmsg) {
  auto fail = [&](int code, const std::string& msg) {
    *errmsg = msg;
    *out = nullptr;
    return code;
  };
  // argument checks: reference Plan::initialize nufft_plan.cc:174-183 / nufft_plan.cu.cc:1833-1843
  if (type != NUFFT_HIP_TYPE_1 && type != NUFFT_HIP_TYPE_2)
    return fail(NUFFT_HIP_UNIMPLEMENTED, "type-3 transforms are not implemented");
  if (rank < 1 || rank > 3)
    return fail(NUFFT_HIP_UNIMPLEMENTED, format("rank %d is not implemented", rank));
  if (ntransf < 1) return fail(NUFFT_HIP_INVALID_ARGUMENT, "num_transforms must be >= 1");
  if (precision != NUFFT_HIP_F32 && precision != NUFFT_HIP_F64)
    return fail(NUFFT_HIP_INVALID_ARGUMENT, "precision must be 4 (float) or 8 (double)");
  if (iflag != NUFFT_HIP_FORWARD && iflag != NUFFT_HIP_BACKWARD)
    return fail(NUFFT_HIP_INVALID_ARGUMENT, "iflag must be -1 (forward) or +1 (backward)");
  if (!grid_dims) return fail(NUFFT_HIP_INVALID_ARGUMENT, "grid_dims is null");
  if (opts_in) {
    // the experiment surface is part of the ABI: unknown bits, contradictory OFF | ON pairs and negative group /
    // lane counts are refused rather than silently resolved
    const int32_t t = opts_in->tuning;
    if (t & ~NUFFT_HIP_TUNE_ALL) return fail(NUFFT_HIP_INVALID_ARGUMENT, format("unknown options.tuning bits 0x%x", (unsigned)(t & ~NUFFT_HIP_TUNE_ALL)));
    static const int pairs[][2] = {{NUFFT_HIP_TUNE_GROUP_OFF, NUFFT_HIP_TUNE_GROUP_ON}, {NUFFT_HIP_TUNE_SPARSE_OFF, NUFFT_HIP_TUNE_SPARSE_ON},
                                   {NUFFT_HIP_TUNE_CELLSORT_OFF, NUFFT_HIP_TUNE_CELLSORT_ON}, {NUFFT_HIP_TUNE_CELLSORT3D_OFF, NUFFT_HIP_TUNE_CELLSORT3D_ON},
                                   {NUFFT_HIP_TUNE_JOINT_OFF, NUFFT_HIP_TUNE_JOINT_ON}, {NUFFT_HIP_TUNE_STAGED_OFF, NUFFT_HIP_TUNE_STAGED_ON},
                                   {NUFFT_HIP_TUNE_SORT2_OFF, NUFFT_HIP_TUNE_SORT2_ON}, {NUFFT_HIP_TUNE_STACK_OFF, NUFFT_HIP_TUNE_STACK_ON},
                                   {NUFFT_HIP_TUNE_ISPLIT_OFF, NUFFT_HIP_TUNE_ISPLIT_ON}, {NUFFT_HIP_TUNE_DIRECT_OFF, NUFFT_HIP_TUNE_DIRECT_ON}};
    for (const auto& pr : pairs)
      if ((t & pr[0]) && (t & pr[1])) return fail(NUFFT_HIP_INVALID_ARGUMENT, format("options.tuning has both bits of an OFF / ON pair (0x%x)", (unsigned)(pr[0] | pr[1])));
    if (opts_in->op_group < 0 || opts_in->op_lanes < 0) return fail(NUFFT_HIP_INVALID_ARGUMENT, "options.op_group and options.op_lanes must be >= 0");
  }

  nufft_hip_plan p = new nufft_hip_plan_s();
  if (opts_in) p->opts = *opts_in; else nufft_hip_default_options(&p->opts);
  p->type = type; p->rank = rank; p->iflag = iflag; p->ntransf = ntransf; p->precision = precision;
  p->stream = (hipStream_t)stream;
  // tolerance clamp: reference nufft_plan.h:84-89, nufft_plan.cu.cc:3062-3064
  const double eps = precision == NUFFT_HIP_F32 ? 6e-8 : 1.1e-16;
  p->tol = std::max(tol, eps);

  Geom& g = p->g;
  std::memset(&g, 0, sizeof(g));
  g.rank = rank;
  p->grid_elems = 1;
  for (int d = 0; d < 3; ++d) {
    g.nmodes[d] = d < rank ? (int)grid_dims[d] : 1;
    if (d < rank && (grid_dims[d] < 1 || grid_dims[d] > kMaxArraySize)) {
      delete p;
      return fail(NUFFT_HIP_INVALID_ARGUMENT, format("invalid grid dimension %lld", (long long)grid_dims[d]));
    }
    p->grid_elems *= g.nmodes[d];
  }

  // Upsampling factor: 2.0 always on the GPU path (reference nufft_plan.cu.cc:1854-1857).
  double sigma = p->opts.upsampling_factor;
  if (sigma == 0.0) sigma = 2.0;
  if (sigma <= 1.0) {
    delete p;
    return fail(NUFFT_HIP_INVALID_ARGUMENT, format("upsampling_factor must be > 1.0, but got: %g", sigma));
  }
  // Kernel width. Reference rule: w = ceil(-log10(tol / 10)) at sigma = 2
  // (nufft_plan.h:762-777), whose value at exact powers of ten depends on the
  // compiler's log10 overload (7 or 8 at tol = 1e-6). This build
  // takes the safe side: one more cell whenever the rule lands within 1e-5 of
  // an integer, so tol = 1e-6 -> 8, 1e-4 -> 6 in either precision.
  int w = p->opts.kernel_width;
  if (w == 0) {
    if (sigma == 2.0) w = (int)std::ceil(-std::log10(p->tol / 10.0) + 1e-5);
    else w = (int)std::ceil(-std::log(p->tol) / (kPi * std::sqrt(1.0 - 1.0 / sigma)) + 1e-5);
    w = std::max(2, std::min(w, kMaxW));
  }
  if (w < 2 || w > kMaxW) {
    delete p;
    return fail(NUFFT_HIP_INVALID_ARGUMENT, format("kernel_width must be in [2, %d], but got: %d", kMaxW, w));
  }
  // ES kernel parameters: reference setup_spreader nufft_plan.cc:923-940, nufft_plan.cu.cc:3078-3093
  double beta_over_w = 2.30;
  if (w == 2) beta_over_w = 2.20;
  if (w == 3) beta_over_w = 2.26;
  if (w == 4) beta_over_w = 2.38;
  if (sigma != 2.0) beta_over_w = 0.97 * kPi * (1.0 - 1.0 / (2.0 * sigma));
  p->ks.w = w; p->ks.sigma = sigma; p->ks.beta = beta_over_w * w; p->ks.c = 4.0 / ((double)w * w);
  g.w = w;

  // Fine grid: reference initialize_fine_grid nufft_plan.h:803-863, set_grid_size nufft_plan.cu.cc:3166-3204
  p->fine_elems = 1;
  for (int d = 0; d < 3; ++d) g.nf[d] = 1;
  for (int d = 0; d < rank; ++d) {
    int64_t n = p->opts.spread_only ? g.nmodes[d] : (int64_t)((double)g.nmodes[d] * sigma);
    if (n < 2 * w) n = 2 * w;
    n = next_smooth_even(n);
    if (p->opts.spread_only && n != g.nmodes[d]) {
      const int bad = g.nmodes[d];
      delete p;
      return fail(NUFFT_HIP_INVALID_ARGUMENT,
                  format("Invalid grid dimension size: %d. Grid dimension must be even, larger than "
                         "the kernel (%d) and have no prime factors larger than 5.", bad, 2 * w));
    }
    g.nf[d] = (int)n;
    p->fine_elems *= n;
  }
  // batch size: reference nufft_plan.cu.cc:1923-1928 (min(ntransf, 8) unless set)
  // (the reference's default of 8 is a memory bound for 16-80 GB cards; with 288 GB of HBM small fine
  // grids take more transforms per launch -- up to 2^23 fine cells per batch (64 MB in float, well
  // inside the 256 MB Infinity Cache; on 1024^2 fine grids 16 or 32 per batch measured level with or
  // behind 8): 16 transforms on a 512^2 fine grid 0.207 -> 0.183 ms (type 1), 0.151 -> 0.127 ms
  // (type 2) -- `max_batch_size` still caps it)
  const int auto_batch = (int)std::max<int64_t>(8, ((int64_t)1 << 23) / std::max<int64_t>(1, p->fine_elems));
  p->batch_size = p->opts.max_batch_size > 0 ? std::min(p->opts.max_batch_size, ntransf)
                                             : std::min(ntransf, auto_batch);
  p->batch_size = std::min(p->batch_size, 32768);   // grid.y limit of the batched launches
  p->nitems = std::max(1, p->opts.num_point_sets);
  if (p->nitems > 1) {
    // every transform of every point set in ONE pass (the kernels index the fine grids by
    // item * ntransf + transform)
    if (p->nitems > 4096 || ntransf > 32768 || (int64_t)p->nitems * ntransf > 65535) {
      delete p;
      return fail(NUFFT_HIP_INVALID_ARGUMENT,
                  "num_point_sets must be <= 4096, num_transforms <= 32768 and their product <= 65535");
    }
    p->batch_size = ntransf;
  }
  if (p->fine_elems * p->batch_size * p->nitems > kMaxArraySize) {
    const long long sz = (long long)(p->fine_elems * p->batch_size * p->nitems);
    delete p;
    return fail(NUFFT_HIP_INVALID_ARGUMENT, format("Fine grid is too big: size %lld > %lld", sz, (long long)kMaxArraySize));
  }

  // Tiles (bins). Defaults: 1024 | 32 x 32 | 16 x 16 x 4 fine cells; shrunk
  // until the LDS tile fits.
  // 3-D: 8 z-planes per tile when the wavefront kernel's LDS planes still fit
  // (w <= 6): fewer halo cells per point (r01: 22.7 -> 21.2 ms at M = 1e8).
  // 2-D float type-2 (and interp-only) plans: 64 x 64. Their LDS tile is single-precision
  // complex (71^2 x 8 B = 40 KB), and four times fewer tiles make the (workgroup, tile) runs
  // of the scatter four times longer (EXPERIMENTS.md section 5: the scatter is transaction bound).
  const bool t2_big = type == NUFFT_HIP_TYPE_2 && rank == 2 && precision == NUFFT_HIP_F32 && w <= 8 &&
                      p->opts.spread_method == NUFFT_HIP_METHOD_AUTO && p->opts.tile_dims[0] == 0 &&
                      p->opts.tile_dims[1] == 0 && p->opts.max_subproblem_size <= 0 &&
                      g.nf[0] >= 64 && g.nf[1] >= 64 &&
                      // (enough of them to fill 256 CUs: a 512^2 fine grid has only 64 -- the reference
                      // benchmark's 256^2 case ran its interp kernel in 30 us on 64 workgroups)
                      (int64_t)g.nf[0] * g.nf[1] >= ((int64_t)1 << 21);
  const int t2d = t2_big ? 64 : 32;
  // 3-D float at w = 8 (fp64 planes, one component per launch): depth 8 as well -- the padded tile
  // (23 x 23 x 15 cells, 66 KB) still lets two workgroups share a CU with a 16-point staging chunk,
  // and the halo written out per fine cell falls from 5.7x to 3.9x.
  // 3-D float at w = 7, 8 (r04): packed fixed point with the exact conversion and the count-filter bound
  // (spread_patch3_kernel, nufft_dense3.hip) on the same depth-8 tiles; lds_accumulate = 1 keeps the fp64 planes
  // (w = 7: on depth-4 tiles, as before), options.tuning FXPATCH_OFF the r03 kernels.
  const bool patch_want = rank == 3 && precision == NUFFT_HIP_F32 && (w == 7 || w == 8) && p->opts.lds_accumulate != 1 &&
                          !(p->opts.tuning & NUFFT_HIP_TUNE_FXPATCH_OFF) &&
                          (p->opts.spread_method == NUFFT_HIP_METHOD_AUTO || p->opts.spread_method == NUFFT_HIP_METHOD_TILE_WAVE);
  const bool deep8 = w <= 6 || (w == 8 && precision == NUFFT_HIP_F32 && (patch_want || p->opts.lds_accumulate != 2)) ||
                     (w == 7 && patch_want);
  int def_tile[3][3] = {{1024, 1, 1}, {t2d, t2d, 1}, {16, 16, deep8 ? 8 : 4}};
  // w = 9..16 (tol < 1e-7): the 16 x 4-lane spread kernels of nufft_wide.hip and their tiles
  // (2-D 32 x 32; 3-D 16 x 8 x 4 up to w = 12, 8 x 8 x 4 above: one fp64 plane of LDS per launch)
  bool wide = !(p->opts.tuning & NUFFT_HIP_TUNE_NO_WIDE) && wide_spread_supported(rank, w) &&
              (p->opts.spread_method == NUFFT_HIP_METHOD_AUTO || p->opts.spread_method == NUFFT_HIP_METHOD_TILE_WAVE) &&
              p->opts.tile_dims[0] == 0 && p->opts.tile_dims[1] == 0 && p->opts.tile_dims[2] == 0;
  int wide_tile[3] = {1, 1, 1};
  if (wide) {
    wide_spread_tile(rank, w, wide_tile);
    for (int d = 0; d < rank; ++d) {
      if (g.nf[d] < wide_tile[d]) wide = false;
      def_tile[rank - 1][d] = wide_tile[d];
    }
  }
  for (int d = 0; d < 3; ++d) {
    int t = d < rank ? (p->opts.tile_dims[d] > 0 ? p->opts.tile_dims[d] : def_tile[rank - 1][d]) : 1;
    t = std::max(1, std::min(t, 1024));
    if (d < rank) t = std::min(t, g.nf[d]);
    if (rank == 3 && precision == NUFFT_HIP_F32) t = std::min(t, 16);   // packed 3-D float records: 4-bit tile-local starts
    g.tile[d] = t;
  }
  const size_t lds_limit = 96 * 1024;
  for (;;) {
    int64_t nt = 1;
    for (int d = 0; d < 3; ++d) {
      g.ntile[d] = d < rank ? (g.nf[d] + g.tile[d] - 1) / g.tile[d] : 1;
      g.ldim[d] = d < rank ? g.tile[d] + w - 1 : 1;
      nt *= g.ntile[d];
    }
    g.nitems = p->nitems;
    g.ntiles_item = (int)nt;
    if (nt * p->nitems > ((int64_t)1 << 30)) {
      delete p;
      return fail(NUFFT_HIP_INVALID_ARGUMENT, "too many tiles (fine grid x num_point_sets)");
    }
    g.ntiles = (int)(nt * p->nitems);
    g.lstride = g.ldim[0];
    for (int d = 0; d < 3; ++d) {
      g.tile_shift[d] = -1;
      for (int b = 0; b <= 10; ++b)
        if (g.tile[d] == (1 << b)) g.tile_shift[d] = b;
    }
    if (wide) break;   // (fixed tiles; the kernel's LDS need is known to fit)
    if (spread_lds_bytes(g, NUFFT_HIP_METHOD_TILE_GENERIC, precision) <= lds_limit) break;
    // the depth-8 tile of the 3-D float wavefront kernel at w = 8 holds one fp64 plane per launch:
    // it fits although two full planes (what the generic kernel would need) do not
    if (rank == 3 && (w == 8 || (w == 7 && patch_want)) && precision == NUFFT_HIP_F32 && g.tile[0] == 16 && g.tile[1] == 16 && g.tile[2] == 8 &&
        (p->opts.spread_method == NUFFT_HIP_METHOD_AUTO || p->opts.spread_method == NUFFT_HIP_METHOD_TILE_WAVE) &&
        (patch_want || p->opts.lds_accumulate != 2))
      break;
    int big = 0;
    for (int d = 1; d < rank; ++d)
      if (g.tile[d] > g.tile[big]) big = d;
    if (g.tile[big] == 1) break;
    g.tile[big] = (g.tile[big] + 1) / 2;
  }
  g.wide = wide ? 1 : 0;
  g.sub_small = 0;
  // Tile numbering. 3-D float plans with more tiles than the LDS-counter sort takes number them by super-tiles of
  // 64^3 fine cells, which the two-level sort (launch_sort, mode 3) sorts by first: every dimension a multiple of
  // 64 cells, at most 4096 super-tiles (r06; 1024 before) of at most 256 tiles each, one point set.
  for (int d = 0; d < 3; ++d) { g.sup_shift[d] = 0; g.nsup[d] = g.ntile[d]; }
  {
    const int mode = tune_mode(g_tuning_probe(p->opts.tuning), NUFFT_HIP_TUNE_SORT2_OFF, NUFFT_HIP_TUNE_SORT2_ON);
    bool ok = rank == 3 && precision == NUFFT_HIP_F32 && p->nitems <= 1 && mode != 0 &&
              (mode > 0 || g.ntiles > kMaxLdsTiles);
    int64_t nsuper = 1;
    int key_bits = 0;
    for (int d = 0; d < 3 && ok; ++d) {
      ok = g.tile_shift[d] >= 0 && g.tile_shift[d] <= 6;
      nsuper *= (g.nf[d] + 63) / 64;
      key_bits += 6 - g.tile_shift[d];
    }
    if (ok && nsuper <= 4096 && key_bits <= 8 && key_bits > 0) {
      // r06: fine grids that are not multiples of 64 cells too (the smooth sizes: 480, 400, 384 ...): the last super-tile of
      // a dimension is partial, and the tile ids of the cells it lacks exist but stay empty -- the tile tables are sized by
      // the padded count (240^3 modes: 65536 ids for 54000 tiles; sort 523 -> see profiles/r06_sort2_partial.txt)
      for (int d = 0; d < 3; ++d) { g.sup_shift[d] = 6 - g.tile_shift[d]; g.nsup[d] = (g.nf[d] + 63) / 64; }
      g.ntiles = g.ntiles_item = (int)(nsuper << key_bits);
    }
  }
  if (!wide && spread_lds_bytes(g, NUFFT_HIP_METHOD_TILE_GENERIC, precision) > 160 * 1024) {
    delete p;
    return fail(NUFFT_HIP_RESOURCE_EXHAUSTED, "kernel too wide for an LDS tile");  // cf. nufft_plan.cu.cc:2458-2463
  }

  // Polynomial kernel table. The fit error falls geometrically with the term
  // count until it reaches a plateau set by the kernel's square-root end-point
  // singularity (~5e-(w+1) of the peak, reached near w + 2 terms); more terms
  // buy nothing. Take the smallest count within 1.3x of that plateau, or below
  // the precision floor.
  p->horner_h.assign(kMaxCoef * kMaxW, 0.0);
  const double floor_err = precision == NUFFT_HIP_F32 ? 2e-8 : 1e-15;
  const double plateau = fit_horner(p->ks, std::min(kMaxCoef, w + 3), p->horner_h.data());
  const double want = std::max(floor_err, 1.3 * plateau);
  int nc = std::max(3, w / 2 + 1);
  for (; nc < kMaxCoef; ++nc)
    if (fit_horner(p->ks, nc, p->horner_h.data()) <= want) break;
  double kmax = 1.0;
  fit_horner(p->ks, nc, p->horner_h.data(), &kmax);
  g.ncoef = nc;
  // Headroom of the packed fixed-point accumulation (3-D float): cell sums are bounded by
  // sum |c| * kmax^rank in exact arithmetic; the float evaluation of the polynomial and of the
  // products adds a few ulp per factor, covered by the 1e-4.
  g.fx_headroom = (float)(std::pow(std::max(1.0, kmax), rank) * (1.0 + 1e-4));

  g.max_sub = p->opts.max_subproblem_size > 0 ? p->opts.max_subproblem_size : 1024;
  bool auto_sub = p->opts.max_subproblem_size <= 0;
  int method = p->opts.spread_method;
  if (method < NUFFT_HIP_METHOD_AUTO || method > NUFFT_HIP_METHOD_POINT_GLOBAL) {
    delete p;
    return fail(NUFFT_HIP_INVALID_ARGUMENT, format("unknown spread_method %d", method));
  }
  // AUTO also lets launch_spread take the LDS-free kernel when the point set turns out sparse
  g.sparse_auto = method == NUFFT_HIP_METHOD_AUTO ? 1 : 0;
  g.fused = 0;
  g.tuning = p->opts.tuning;
  const bool t2_wave = t2_big && g.tile[0] == 64 && g.tile[1] == 64;   // (not shrunk by a tiny grid)
  if (method == NUFFT_HIP_METHOD_AUTO)
    method = (wide || t2_wave || wave_method_supported(g, precision)) ? NUFFT_HIP_METHOD_TILE_WAVE : NUFFT_HIP_METHOD_TILE_GENERIC;
  if (method != NUFFT_HIP_METHOD_TILE_WAVE) g.wide = 0;
  // 1-D plans: interp_line_kernel (nufft_line.hip) behind the automatic choice (an explicit
  // TILE_GENERIC keeps the gather-from-global kernel: the tests' second opinion)
  g.line = (!(p->opts.tuning & NUFFT_HIP_TUNE_NO_LINE) && p->opts.spread_method == NUFFT_HIP_METHOD_AUTO && method == NUFFT_HIP_METHOD_TILE_GENERIC &&
            line_kernels_supported(g)) ? 1 : 0;
  // (type 2: one tile load per 4096 points; a spread_only plan serves both ops and keeps 1024)
  if (g.line && auto_sub && type == NUFFT_HIP_TYPE_2 && !p->opts.spread_only) g.max_sub = 4096;
  if (method == NUFFT_HIP_METHOD_TILE_WAVE && !wide && !t2_wave && !wave_method_supported(g, precision)) {
    delete p;
    return fail(NUFFT_HIP_INVALID_ARGUMENT,
                "spread_method TILE_WAVE needs rank 2 or 3 and the default tile sizes");
  }
  p->method = method;
  // packed 32+32-bit fixed-point accumulation: the 3-D float wavefront kernel at w <= 7.
  // The quantisation step is (sum of |c| over the SUBPROBLEM) / 2^31 against a typical
  // contribution of |c| * 0.04 (w = 7-8 kernel products), i.e. a relative error of about
  // 4e-9 * points per subproblem: w = 7 (tol 1e-5) caps the subproblem at 512 points
  // (measured +1.4e-6); w = 8 (tol 1e-6) would need ~25 and keeps the fp64 planes
  // (measured 2.4e-6 at 256 points per subproblem).
  g.fixed_point = 0;
  g.split_reim = 0;
  g.cell_sorted = 0;
  g.stack = 0;
  g.stack_len = g.stack_cap = 0;
  g.fx_patch = (patch_want && method == NUFFT_HIP_METHOD_TILE_WAVE && !g.wide && g.tile[0] == 16 && g.tile[1] == 16 &&
                g.tile[2] == 8) ? 1 : 0;
  if (method == NUFFT_HIP_METHOD_TILE_WAVE && rank == 3 && precision == NUFFT_HIP_F32 && (w <= 7 || g.fx_patch) &&
      p->opts.lds_accumulate != 1 && (w <= 6 || g.fx_patch || g.tile[2] == 4))
    g.fixed_point = 1;
  g.fp64_stack = (method == NUFFT_HIP_METHOD_TILE_WAVE && rank == 3 && precision == NUFFT_HIP_F64 && !g.wide && w >= 2 && w <= 8 &&
                  g.tile[0] == 16 && g.tile[1] == 16 && g.tile[2] == (w <= 6 ? 8 : 4)) ? 1 : 0;
  if (method == NUFFT_HIP_METHOD_TILE_WAVE && rank == 3 && g.wide) {   // w = 9..16 (either precision): spread_wide_kernel<..., STACK>
    int wt[3];
    wide_spread_tile(3, w, wt);
    g.fp64_stack = (g.tile[0] == wt[0] && g.tile[1] == wt[1] && g.tile[2] == wt[2]) ? 1 : 0;
  }
  if (p->opts.lds_accumulate == 2 && !g.fixed_point) {
    delete p;
    return fail(NUFFT_HIP_INVALID_ARGUMENT,
                "lds_accumulate = 2 (fixed point) needs the 3-D float wavefront method with kernel width <= 8");
  }
  // The output error the quantisation adds is <= ~3.5e-9 B for uniform strengths (measured r04 at w = 8: 0.49e-7 in
  // quadrature at a mean bound of 14, 0.61e-7 at 28; r02 / r03: 2.4e-6 at a step of 2^-22 largest strengths = B of
  // 512); a subproblem may spend 0.28 of the width's tolerance 10^(2 - w) on it: B <= 80 at w = 8.
  g.fx_bound_limit = (float)(0.28 * std::pow(10.0, 2 - w) / 3.5e-9);
#ifdef NUFFT_FX_BOUND_LIMIT   // experiment builds (tools/fx_error_vs_crest.sh): a fixed limit instead of the rule
  g.fx_bound_limit = (float)(NUFFT_FX_BOUND_LIMIT);
#endif
  if (g.fx_patch) {
    // per-tap maxima of the fitted polynomials over z in [-1, 1] (sampled; margins for the sampling, the float
    // evaluation and the products)
    for (int t = 0; t < 16; ++t) p->taps.k[t] = 0.f;
    for (int t = 0; t < w; ++t) {
      double m = 0.0;
      for (int sidx = 0; sidx <= 4096; ++sidx) {
        const double z = -1.0 + 2.0 * sidx / 4096.0;
        double acc = p->horner_h[(nc - 1) * kMaxW + t];
        for (int k = nc - 2; k >= 0; --k) acc = acc * z + p->horner_h[k * kMaxW + t];
        m = std::max(m, std::fabs(acc));
      }
      p->taps.k[t] = (float)(m * 1.001 + 1e-6);
    }
    for (int t = 0; t < 8; ++t) g.fx_tap[t] = p->taps.k[t];
  }
  // fp64 planes in 3-D float at tile depth 4 (w = 8, or w <= 7 with lds_accumulate = 1):
  // one plane per launch, so that two workgroups share a CU (EXPERIMENTS.md section 4)
  g.split_reim = (method == NUFFT_HIP_METHOD_TILE_WAVE && rank == 3 && precision == NUFFT_HIP_F32 &&
                  !g.fixed_point && (g.tile[2] == 4 || (g.tile[2] == 8 && w == 8))) ? 1 : 0;
  if (method == NUFFT_HIP_METHOD_TILE_WAVE)
    g.lstride = g.wide ? wide_spread_lstride(rank, w) : (t2_wave ? 72 : wave_lstride(rank));
  // wavefront kernels: one subproblem per typical tile measured fastest (r01 sweeps);
  // every extra subproblem of a tile repeats its zero-fill and write-out
  if (method == NUFFT_HIP_METHOD_TILE_WAVE && auto_sub) g.max_sub = t2_wave ? 16384 : 4096;   // (interp: one tile load per subproblem)
  // clustered point sets: scan_tiles_kernel falls back to 4096-point subproblems
  if (method == NUFFT_HIP_METHOD_TILE_WAVE && auto_sub && t2_wave) g.sub_small = 4096;
  if (g.wide && auto_sub && rank == 3) g.max_sub = 1024;   // (hundreds of LDS atomics per point: keep workgroups short)
  const bool spreads = type == NUFFT_HIP_TYPE_1 || p->opts.spread_only;
  if (g.fixed_point && w > 6 && !g.fx_patch) g.max_sub = std::min(g.max_sub, 512);
  // w = 8: 2560 points on 2048 cells keep the bound of a uniform subproblem at 55-65 of the 80 tol 1e-6 allows
  // (and tiles of config 4's density, 1526 points on average, in ONE subproblem: a cap of 1536 split 40 % of them)
  if (g.fx_patch && w == 8 && auto_sub && spreads) g.max_sub = 2560;
  g.fx_max_subs = 16;
  g.fold_pow2 = (p->opts.tuning & NUFFT_HIP_TUNE_QFOLD_OFF) ? 0 : 1;
  for (int d = 0; d < rank; ++d)
    if (g.tile_shift[d] < 0) g.fold_pow2 = 0;
  p->lds_bytes = spread_lds_bytes(g, method, precision);
  if (p->lds_bytes > 160 * 1024 || interp_lds_bytes(g, method, precision) > 160 * 1024) {
    // does not fit (e.g. 3-D double at w = 8): fall back to the generic tile path
    if (p->opts.spread_method == NUFFT_HIP_METHOD_TILE_WAVE) {
      delete p;
      return fail(NUFFT_HIP_RESOURCE_EXHAUSTED, "spread_method TILE_WAVE does not fit in LDS for this configuration");
    }
    p->method = method = NUFFT_HIP_METHOD_TILE_GENERIC;
    g.lstride = g.ldim[0];
    p->lds_bytes = spread_lds_bytes(g, method, precision);
  }

  for (int d = 0; d < rank; ++d) kernel_fseries(p->ks, g.nf[d], p->fser_h[d]);
  p->spread_scale = p->opts.spread_only ? spread_only_scale(p->ks, rank) : 1.0;

  *out = p;
  return NUFFT_HIP_OK;
}

static int create_common(nufft_hip_plan* out, int type, int rank, const int64_t* grid_dims,
                         int iflag, int ntransf, double tol, int precision,
                         const nufft_hip_options* opts_in, void* stream,
                         const nufft_hip_allocator* allocator, bool host_only, char* errbuf,
                         size_t errbuf_len) {
  auto fail = [&](int code, const std::string& msg) {
    if (errbuf && errbuf_len) snprintf(errbuf, errbuf_len, "%s", msg.c_str());
    if (out) *out = nullptr;
    return code;
  };
  if (!out) return NUFFT_HIP_INVALID_ARGUMENT;
  nufft_hip_plan p = nullptr;
  std::string msg;
  int rc0 = configure(&p, type, rank, grid_dims, iflag, ntransf, tol, precision, opts_in, stream, &msg);
  if (rc0) return fail(rc0, msg);
  if (host_only) {
    p->host_only = true;
    *out = p;
    return NUFFT_HIP_OK;
  }
  if (allocator && allocator->alloc) p->allocator = *allocator;
  if (hipGetDevice(&p->device) != hipSuccess) {
    delete p;
    (void)hipGetLastError();
    return fail(NUFFT_HIP_INTERNAL, "no HIP device available (this library has no CPU fallback)");
  }
  // device code first (see preload_device_code), then device state
  if (const hipError_t pe = preload_device_code(); pe != hipSuccess) {
    const std::string m = format("loading the device code failed: %s", hipGetErrorString(pe));
    (void)hipGetLastError();
    delete p;
    return fail(NUFFT_HIP_INTERNAL, m);
  }
  p->own_fft = !p->opts.spread_only && pruned_fft_supported(p->g, precision);
  int rc = precision == NUFFT_HIP_F32 ? upload_tables<float>(p) : upload_tables<double>(p);
  if (!rc && !p->opts.spread_only && !p->own_fft) {
    // the FFT of full batches and of the remainder batch, so that execute never plans
    rc = build_fft_plan(p, p->batch_size * p->nitems);
    if (!rc && p->nitems == 1) rc = build_fft_plan(p, p->ntransf % p->batch_size);
  }
  // internal allocation: the fixed workspace now, so that nothing allocates after warm-up;
  // a framework allocator is asked at the first set_points (its memory is per call)
  if (!rc && !p->allocator.alloc) rc = ensure_fixed_workspace(p);
  if (rc) {
    const std::string msg2 = p->err;
    destroy(p);
    return fail(rc, msg2);
  }
  *out = p;
  return NUFFT_HIP_OK;
}

int nufft_hip_plan_create(nufft_hip_plan* out, int type, int rank, const int64_t* grid_dims,
                          int iflag, int ntransf, double tol, int precision,
                          const nufft_hip_options* opts_in, void* stream, char* errbuf,
                          size_t errbuf_len) {
  return create_common(out, type, rank, grid_dims, iflag, ntransf, tol, precision, opts_in, stream,
                       nullptr, false, errbuf, errbuf_len);
}

int nufft_hip_plan_create_ex(nufft_hip_plan* out, int type, int rank, const int64_t* grid_dims,
                             int iflag, int ntransf, double tol, int precision,
                             const nufft_hip_options* opts_in, void* stream,
                             const nufft_hip_allocator* allocator, char* errbuf, size_t errbuf_len) {
  return create_common(out, type, rank, grid_dims, iflag, ntransf, tol, precision, opts_in, stream,
                       allocator, false, errbuf, errbuf_len);
}

int nufft_hip_plan_create_host(nufft_hip_plan* out, int type, int rank, const int64_t* grid_dims,
                               int iflag, int ntransf, double tol, int precision,
                               const nufft_hip_options* opts_in, char* errbuf, size_t errbuf_len) {
  return create_common(out, type, rank, grid_dims, iflag, ntransf, tol, precision, opts_in, nullptr,
                       nullptr, true, errbuf, errbuf_len);
}

int nufft_hip_plan_release_workspace(nufft_hip_plan p) {
  if (!p || p->host_only) return NUFFT_HIP_INVALID_ARGUMENT;
  if (!p->allocator.alloc) HIP_TRY(p, hipStreamSynchronize(p->stream));
  release_workspace(p);
  return NUFFT_HIP_OK;
}

int nufft_hip_plan_set_allocator(nufft_hip_plan p, const nufft_hip_allocator* allocator) {
  if (!p || p->host_only) return NUFFT_HIP_INVALID_ARGUMENT;
  if (p->fixed_ws || p->rec || p->rec2 || p->hist || p->tile_of) {
    p->err = "set_allocator: the plan still holds workspace (call nufft_hip_plan_release_workspace first)";
    return NUFFT_HIP_INVALID_ARGUMENT;
  }
  if (allocator && allocator->alloc) p->allocator = *allocator;
  else p->allocator = {nullptr, nullptr, nullptr};
  return NUFFT_HIP_OK;
}

int nufft_hip_plan_describe(int type, int rank, const int64_t* grid_dims, int iflag, int ntransf,
                            double tol, int precision, const nufft_hip_options* opts,
                            nufft_hip_plan_info* info, char* errbuf, size_t errbuf_len) {
  nufft_hip_plan p = nullptr;
  std::string msg;
  int rc = configure(&p, type, rank, grid_dims, iflag, ntransf, tol, precision, opts, nullptr, &msg);
  if (rc) {
    if (errbuf && errbuf_len) snprintf(errbuf, errbuf_len, "%s", msg.c_str());
    return rc;
  }
  rc = nufft_hip_plan_get_info(p, info);
  delete p;
  return rc;
}

int nufft_hip_set_points(nufft_hip_plan p, int64_t M, const void* x, const void* y, const void* z,
                         int64_t stride) {
  if (!p) return NUFFT_HIP_INVALID_ARGUMENT;
  if (p->host_only) {
    p->err = "this plan was created host-only (no device state)";
    return NUFFT_HIP_INVALID_ARGUMENT;
  }
  // (tile_start / sub_start, the scatter cursors and the subproblem search are 32-bit over ALL sets)
  if (M < 0 || M > kMaxArraySize || M * std::max(1, p->nitems) > kMaxArraySize) {
    p->err = format("invalid number of points %lld (x %d point sets; at most %lld in total)", (long long)M,
                    std::max(1, p->nitems), (long long)kMaxArraySize);
    return NUFFT_HIP_INVALID_ARGUMENT;
  }
  if (M > 0 && (!x || (p->rank > 1 && !y) || (p->rank > 2 && !z))) {
    p->err = "null points pointer";
    return NUFFT_HIP_INVALID_ARGUMENT;
  }
  if (stride < 1) stride = 1;
  return p->precision == NUFFT_HIP_F32 ? set_points_impl<float>(p, M, x, y, z, stride)
                                       : set_points_impl<double>(p, M, x, y, z, stride);
}

#define NUFFT_REQUIRE_DEVICE_PLAN(p)                                        \
  do {                                                                      \
    if (!(p)) return NUFFT_HIP_INVALID_ARGUMENT;                            \
    if ((p)->host_only) {                                                   \
      (p)->err = "this plan was created host-only (no device state)";      \
      return NUFFT_HIP_INVALID_ARGUMENT;                                    \
    }                                                                       \
  } while (0)

// r06, direct type-2 interpolation (tools/exp_direct_interp.py, profiles/r06_direct_interp.txt: us per tfft.nufft call, direct /
// sorted): 2-D 256^2 modes M = 2e4 32 / 39, 1e5 35 / 41, 2e5 44 / 45, 5e5 65 / 56; 1024^2: 1e5 60 / 72, 2e5 82 / 74; 3-D 128^3: 2e4 153 / 270,
// 1e5 267 / 268, 2e5 469 / 278; 64^3: 2e4 80 / 67 -- in 2-D up to 1e5 points, in 3-D while a point has > 256 fine cells to itself
constexpr int64_t kDirectMaxPoints2d = 100000;
constexpr int64_t kDirectCellsPerPoint3d = 256;

int nufft_hip_execute_with_points(nufft_hip_plan p, int64_t M, const void* x, const void* y,
                                  const void* z, int64_t stride, void* c, void* f) {
  NUFFT_REQUIRE_DEVICE_PLAN(p);
  // (tile_start / sub_start, the scatter cursors and the subproblem search are 32-bit over ALL sets)
  if (M < 0 || M > kMaxArraySize || M * std::max(1, p->nitems) > kMaxArraySize) {
    p->err = format("invalid number of points %lld (x %d point sets; at most %lld in total)", (long long)M,
                    std::max(1, p->nitems), (long long)kMaxArraySize);
    return NUFFT_HIP_INVALID_ARGUMENT;
  }
  if (M > 0 && (!x || (p->rank > 1 && !y) || (p->rank > 2 && !z))) {
    p->err = "null points pointer";
    return NUFFT_HIP_INVALID_ARGUMENT;
  }
  if (stride < 1) stride = 1;
  const bool type1 = p->type == NUFFT_HIP_TYPE_1;
  // one type-1 transform: the sort can carry the strengths inside the records
  const bool fuse = type1 && p->ntransf == 1 && p->precision == NUFFT_HIP_F32 && c &&
                    fused_sort_supported(p->g, p->method, p->precision, M);
  // r06: a small type-2 call interpolates straight from these points -- no sort (four launches) for points that are
  // used once
  if (!type1 && !p->opts.spread_only && M > 0 && direct_interp_supported(p->g) && p->stop_after < 0 &&
      !(p->opts.check_points_range && p->opts.points_range != NUFFT_HIP_RANGE_INFINITE) &&
      p->opts.spread_method == NUFFT_HIP_METHOD_AUTO) {
    const int mode = tune_mode(p->g, NUFFT_HIP_TUNE_DIRECT_OFF, NUFFT_HIP_TUNE_DIRECT_ON);
    const bool small = p->rank == 2 ? M <= kDirectMaxPoints2d : M * kDirectCellsPerPoint3d <= (int64_t)p->fine_elems;
    if (mode > 0 || (mode < 0 && small)) {
      int rc = ensure_fixed_workspace(p);
      if (rc) return rc;
      PointsIn in;
      in.pts[0] = x; in.pts[1] = p->rank > 1 ? y : x; in.pts[2] = p->rank > 2 ? z : x;
      in.stride = stride;
      in.M = in.M_item = M;
      in.blocks_per_item = 1;
      in.range_mode = p->opts.points_range;
      in.check_range = 0;
      in.strengths = nullptr;
      in.aos = 0;
      p->M = M;
      p->points_set = false;   // (nothing of these points stays in the plan)
      p->direct_in = &in;
      rc = p->precision == NUFFT_HIP_F32 ? execute_impl<float>(p, c, f) : execute_impl<double>(p, c, f);
      p->direct_in = nullptr;
      return rc;
    }
  }
  int rc = p->precision == NUFFT_HIP_F32 ? set_points_impl<float>(p, M, x, y, z, stride, fuse ? c : nullptr)
                                         : set_points_impl<double>(p, M, x, y, z, stride);
  if (rc) return rc;
  if (p->opts.spread_only)
    rc = p->precision == NUFFT_HIP_F32 ? spread_interp_impl<float>(p, type1 ? 1 : 2, c, f)
                                       : spread_interp_impl<double>(p, type1 ? 1 : 2, c, f);
  else
    rc = p->precision == NUFFT_HIP_F32 ? execute_impl<float>(p, c, f) : execute_impl<double>(p, c, f);
  if (fuse) {   // the records hold THESE strengths: the points are consumed
    p->points_set = false;
    p->fused = false;
    p->g.fused = 0;
  }
  return rc;
}

int nufft_hip_debug_sort_path(nufft_hip_plan p) {
  if (!p || !p->points_set) return -1;
  return sort_mode(p->g, p->M * p->nitems);
}

int64_t nufft_hip_debug_sub_bounds(nufft_hip_plan p, float* out, int64_t n) {
  if (!p || p->host_only || !p->points_set) return -1;
  if (!p->g.fx_patch || !p->sub_bound) return 0;
  if (p->g.stack) {   // one bound per stack
    int32_t live = 0;
    if (hipMemcpyAsync(&live, p->segs, sizeof(int32_t), hipMemcpyDeviceToHost, p->stream) != hipSuccess ||
        hipStreamSynchronize(p->stream) != hipSuccess)
      return -1;
    const int64_t have = std::min<int64_t>(live, (int64_t)stack_grid_bound(p->g, p->M * p->nitems));
    const int64_t take = std::min(n, have);
    if (take > 0 && out &&
        (hipMemcpyAsync(out, p->sub_bound, sizeof(float) * (size_t)take, hipMemcpyDeviceToHost, p->stream) != hipSuccess ||
         hipStreamSynchronize(p->stream) != hipSuccess))
      return -1;
    return have;
  }
  const int64_t slots = (int64_t)subproblem_grid_bound(p->g, p->M * p->nitems);
  const int64_t take = std::min(n, slots);
  if (take > 0 && out) {
    // (launch slots past the live subproblems are never written: report them as 0)
    std::vector<int32_t> ss(1);
    if (hipMemcpyAsync(ss.data(), p->sub_start + p->g.ntiles, sizeof(int32_t), hipMemcpyDeviceToHost, p->stream) != hipSuccess ||
        hipMemcpyAsync(out, p->sub_bound, sizeof(float) * (size_t)take, hipMemcpyDeviceToHost, p->stream) != hipSuccess ||
        hipStreamSynchronize(p->stream) != hipSuccess)
      return -1;
    for (int64_t i = ss[0]; i < take; ++i) out[i] = 0.f;
  }
  return slots;
}

int nufft_hip_debug_stack_params(nufft_hip_plan p, int len, int cap) {
  if (!p || p->host_only || len < 0 || cap < 0 || len > 32767) return NUFFT_HIP_INVALID_ARGUMENT;
  p->g.stack_len = len;
  p->g.stack_cap = cap;
  p->points_set = false;   // (the next set_points cuts by the new rule; its workspace is sized for it)
  return NUFFT_HIP_OK;
}

int64_t nufft_hip_debug_stacks(nufft_hip_plan p, int32_t* out, int64_t n) {
  if (!p || p->host_only || !p->points_set) return -1;
  if (!p->g.stack || !p->segs) return 0;
  int32_t live = 0;
  if (hipMemcpyAsync(&live, p->segs, sizeof(int32_t), hipMemcpyDeviceToHost, p->stream) != hipSuccess ||
      hipStreamSynchronize(p->stream) != hipSuccess)
    return -1;
  const int64_t have = std::min<int64_t>(live, (int64_t)stack_grid_bound(p->g, p->M * p->nitems));
  const int64_t take = std::min(n, have);
  if (take > 0 && out &&
      (hipMemcpyAsync(out, p->segs + 1, sizeof(int4) * (size_t)take, hipMemcpyDeviceToHost, p->stream) != hipSuccess ||
       hipStreamSynchronize(p->stream) != hipSuccess))
    return -1;
  return have;
}

int nufft_hip_debug_shader_clock_mhz(void* stream, double* mhz) {
  if (!mhz) return NUFFT_HIP_INVALID_ARGUMENT;
  if (preload_device_code() != hipSuccess) return NUFFT_HIP_INTERNAL;
  unsigned long long* scratch = nullptr;
  if (hipMalloc((void**)&scratch, sizeof(unsigned long long) * 2 * 512) != hipSuccess) return NUFFT_HIP_RESOURCE_EXHAUSTED;
  const hipError_t e = measure_shader_clock_mhz((hipStream_t)stream, scratch, mhz);
  (void)hipFree(scratch);
  return e == hipSuccess ? NUFFT_HIP_OK : NUFFT_HIP_INTERNAL;
}

int nufft_hip_debug_stop_after(nufft_hip_plan p, int stage) {
  if (!p) return NUFFT_HIP_INVALID_ARGUMENT;
  p->stop_after = stage;
  if (!p->host_only && p->own_fft && !p->opts.spread_only && (stage == STAGE_FFT || stage == STAGE_DECONVOLVE)) {
    // the fine grid between FFT and deconvolution only exists on the rocFFT path: plan it now
    // (debug only: this allocates and synchronises)
    int rc = build_fft_plan(p, p->batch_size * p->nitems);
    if (!rc && p->nitems == 1) rc = build_fft_plan(p, p->ntransf % p->batch_size);
    if (rc) return rc;
    // (a plan whose fixed workspace is not allocated yet gets the buffer from ensure_fixed_workspace)
    if (p->fft_work_bytes && !p->fft_work && p->fixed_ws) {
      if ((rc = dev_alloc(p, &p->fft_work, p->fft_work_bytes))) return rc;
      FFT_TRY(p, rocfft_execution_info_set_work_buffer(p->fft_info, p->fft_work, p->fft_work_bytes));
    }
  }
  return NUFFT_HIP_OK;
}

int nufft_hip_execute(nufft_hip_plan p, void* c, void* f) {
  NUFFT_REQUIRE_DEVICE_PLAN(p);
  if (p->opts.spread_only) {
    p->err = "execute is not available on a spread_only plan";
    return NUFFT_HIP_INVALID_ARGUMENT;
  }
  return p->precision == NUFFT_HIP_F32 ? execute_impl<float>(p, c, f) : execute_impl<double>(p, c, f);
}

int nufft_hip_spread(nufft_hip_plan p, const void* c, void* f) {
  NUFFT_REQUIRE_DEVICE_PLAN(p);
  return p->precision == NUFFT_HIP_F32 ? spread_interp_impl<float>(p, 1, const_cast<void*>(c), f)
                                       : spread_interp_impl<double>(p, 1, const_cast<void*>(c), f);
}

int nufft_hip_interp(nufft_hip_plan p, void* c, const void* f) {
  NUFFT_REQUIRE_DEVICE_PLAN(p);
  return p->precision == NUFFT_HIP_F32 ? spread_interp_impl<float>(p, 2, c, const_cast<void*>(f))
                                       : spread_interp_impl<double>(p, 2, c, const_cast<void*>(f));
}

int nufft_hip_plan_get_info(nufft_hip_plan p, nufft_hip_plan_info* info) {
  if (!p || !info) return NUFFT_HIP_INVALID_ARGUMENT;
  std::memset(info, 0, sizeof(*info));
  info->type = p->type; info->rank = p->rank; info->precision = p->precision; info->iflag = p->iflag;
  info->ntransf = p->ntransf; info->batch_size = p->batch_size;
  info->kernel_width = p->g.w; info->ncoef = p->g.ncoef; info->spread_method = p->method;
  info->upsampling_factor = p->ks.sigma; info->beta = p->ks.beta; info->tol = p->tol;
  for (int d = 0; d < 3; ++d) {
    info->grid_dims[d] = p->g.nmodes[d]; info->fine_dims[d] = p->g.nf[d];
    info->tile_dims[d] = p->g.tile[d]; info->num_tiles[d] = p->g.ntile[d];
  }
  info->max_subproblem_size = p->g.max_sub;
  info->num_points = p->M;
  info->workspace_bytes = p->workspace_bytes;
  return NUFFT_HIP_OK;
}

int nufft_hip_plan_set_stream(nufft_hip_plan p, void* stream) {
  if (!p || p->host_only) return NUFFT_HIP_INVALID_ARGUMENT;
  if (p->stream != (hipStream_t)stream) p->fine_clear_slots = 0;   // the old stream may still be zeroing the fine grid
  p->stream = (hipStream_t)stream;
  if (p->fft_info) FFT_TRY(p, rocfft_execution_info_set_stream(p->fft_info, p->stream));
  return NUFFT_HIP_OK;
}

int nufft_hip_plan_set_timing(nufft_hip_plan p, int enable) {
  if (!p) return NUFFT_HIP_INVALID_ARGUMENT;
  p->timing = enable;
  return NUFFT_HIP_OK;
}

int nufft_hip_plan_get_timing(nufft_hip_plan p, double* ms, int32_t* calls, int n) {
  if (!p || p->host_only) return NUFFT_HIP_INVALID_ARGUMENT;
  HIP_TRY(p, hipStreamSynchronize(p->stream));
  for (auto& pe : p->pending) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, pe.e0, pe.e1) == hipSuccess) {
      p->stage_ms[pe.stage] += t;
      p->stage_calls[pe.stage] += 1;
    }
    p->free_events.push_back(pe.e0);
    p->free_events.push_back(pe.e1);
  }
  p->pending.clear();
  for (int s2 = 0; s2 < n && s2 < STAGE_COUNT; ++s2) {
    if (ms) ms[s2] = p->stage_ms[s2];
    if (calls) calls[s2] = p->stage_calls[s2];
  }
  for (int s2 = 0; s2 < STAGE_COUNT; ++s2) { p->stage_ms[s2] = 0; p->stage_calls[s2] = 0; }
  return NUFFT_HIP_OK;
}

const char* nufft_hip_last_error(nufft_hip_plan p) { return p ? p->err.c_str() : "null plan"; }

int nufft_hip_plan_destroy(nufft_hip_plan p) {
  destroy(p);
  return NUFFT_HIP_OK;
}

int nufft_hip_debug_fine_grid(nufft_hip_plan p, void** fine, int64_t* count) {
  if (!p || !fine || !count) return NUFFT_HIP_INVALID_ARGUMENT;
  *fine = p->d_fine;
  *count = p->fine_elems * p->batch_size * p->nitems;
  return NUFFT_HIP_OK;
}

int nufft_hip_debug_copy_fine_grid(nufft_hip_plan p, void* dst, int64_t count) {
  if (!p || p->host_only || !dst || !p->d_fine || count < 0 || count > p->fine_elems * p->batch_size * p->nitems)
    return NUFFT_HIP_INVALID_ARGUMENT;
  HIP_TRY(p, hipMemcpyAsync(dst, p->d_fine, (size_t)count * 2 * (size_t)p->precision, hipMemcpyDeviceToDevice,
                            p->stream));
  return NUFFT_HIP_OK;
}

int nufft_hip_debug_fseries(nufft_hip_plan p, int dim, double* out, int64_t n) {
  if (!p || dim < 0 || dim >= p->rank || !out) return NUFFT_HIP_INVALID_ARGUMENT;
  const int64_t m = std::min<int64_t>(n, (int64_t)p->fser_h[dim].size());
  for (int64_t k = 0; k < m; ++k) out[k] = p->fser_h[dim][k];
  return NUFFT_HIP_OK;
}

int nufft_hip_debug_eval_kernel(nufft_hip_plan p, int n, const double* x1, double* out) {
  if (!p || !x1 || !out) return NUFFT_HIP_INVALID_ARGUMENT;
  const int w = p->g.w, nc = p->g.ncoef;
  for (int i = 0; i < n; ++i) {
    const double z = 2.0 * x1[i] + w - 1.0;
    for (int j = 0; j < w; ++j) {
      double acc = p->horner_h[(nc - 1) * kMaxW + j];
      for (int k = nc - 2; k >= 0; --k) acc = acc * z + p->horner_h[k * kMaxW + j];
      out[(size_t)i * w + j] = acc;
    }
  }
  return NUFFT_HIP_OK;
}

}  // extern "C"
