// oracle/ref_harness.cc -- thin C entry points around the two pieces of the
// reference that compile standalone. TEST INFRASTRUCTURE ONLY.
//
// Built by oracle/Makefile ONLY when /root/reference is present, from the
// reference sources where they lie (-I/root/reference/...); outputs go to
// oracle/_ref/ (git-ignored). Nothing of the reference is copied into this
// repository: this file only #includes it at build time, the same way
// tensorflow_nufft/cc/kernels/nufft_plan.cc:1291-1307 (eval_kernel_vec_Horner)
// includes its generated tables, and nufft_util.cc:85 calls
// legendre_compute_glr.
#include <cstdio>
#include "tensorflow_nufft/cc/kernels/legendre_rule_fast.h"

namespace {
template <typename FloatType>
void ref_eval_horner(FloatType* ker, const FloatType x, const int w,
                     const double upsampling_factor) {
  FloatType z = 2 * x + w - 1.0;  // nufft_plan.cc:1299
  if (upsampling_factor == 2.0) {
#include "tensorflow_nufft/cc/kernels/kernel_horner_sigma2.inc"
  } else if (upsampling_factor == 1.25) {
#include "tensorflow_nufft/cc/kernels/kernel_horner_sigma125.inc"
  }
}
template <typename FloatType>
void ref_eval_horner_gpu(FloatType* ker, const FloatType x, const int w) {
  FloatType z = 2 * x + w - 1.0;  // nufft_plan.cu.cc:454-462
#include "tensorflow_nufft/cc/kernels/kernel_horner_sigma2_gpu.inc"
}
}  // namespace

extern "C" {
// ker must hold 16 + 4 values per point (the CPU tables are padded to 4).
void ref_horner_f64(int n, const double* x1, int w, double sigma, double* ker,
                    int ld) {
  for (int i = 0; i < n; ++i) ref_eval_horner<double>(ker + (size_t)i * ld, x1[i], w, sigma);
}
void ref_horner_f32(int n, const float* x1, int w, double sigma, float* ker,
                    int ld) {
  for (int i = 0; i < n; ++i) ref_eval_horner<float>(ker + (size_t)i * ld, x1[i], w, sigma);
}
void ref_horner_gpu_f64(int n, const double* x1, int w, double* ker, int ld) {
  for (int i = 0; i < n; ++i) ref_eval_horner_gpu<double>(ker + (size_t)i * ld, x1[i], w);
}
void ref_legendre_glr(int n, double* x, double* w) { legendre_compute_glr(n, x, w); }
}
