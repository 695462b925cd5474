// Standalone C++ client of the C ABI (no Python, no torch, no TensorFlow):
// allocates device buffers with the HIP runtime, runs a type-1 and a type-2
// transform through include/nufft_hip.h and checks them against a direct
// O(M N) sum computed on the host in double precision.
//
//   make -C tensorflow-nufft_amd/csrc examples && ./tensorflow-nufft_amd/csrc/examples/abi_client
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <random>
#include <vector>

#include "nufft_hip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_NUFFT(x) do { int r_ = (x); if (r_) { printf("nufft_hip error %d: %s\n", r_, plan ? nufft_hip_last_error(plan) : err); return 3; } } while (0)

int main() {
  const int N1 = 20, N2 = 24;   // x fastest: array is [N2][N1]
  const int64_t M = 3000;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> up(-3.14159265f, 3.14159265f), uc(-0.5f, 0.5f);
  std::vector<float> pts(2 * M);                 // [M][2] = (y, x) per point, TF order: last coordinate is x
  std::vector<std::complex<float>> c(M), f((size_t)N1 * N2);
  for (auto& v : pts) v = up(rng);
  for (auto& v : c) v = {uc(rng), uc(rng)};

  float *d_pts = nullptr, *d_c = nullptr, *d_f = nullptr;
  CHECK_HIP(hipMalloc(&d_pts, pts.size() * sizeof(float)));
  CHECK_HIP(hipMalloc(&d_c, c.size() * sizeof(float) * 2));
  CHECK_HIP(hipMalloc(&d_f, f.size() * sizeof(float) * 2));
  CHECK_HIP(hipMemcpy(d_pts, pts.data(), pts.size() * sizeof(float), hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_c, c.data(), c.size() * sizeof(float) * 2, hipMemcpyHostToDevice));

  char err[512] = {0};
  nufft_hip_plan plan = nullptr;
  nufft_hip_options opts;
  nufft_hip_default_options(&opts);
  const int64_t dims[3] = {N1, N2, 1};
  // type 1, forward: f[k2][k1] = sum_j c_j exp(-i (k1 x_j + k2 y_j))
  CHECK_NUFFT(nufft_hip_plan_create(&plan, NUFFT_HIP_TYPE_1, 2, dims, NUFFT_HIP_FORWARD, 1, 1e-6, NUFFT_HIP_F32,
                                    &opts, nullptr, err, sizeof err));
  CHECK_NUFFT(nufft_hip_set_points(plan, M, d_pts + 1, d_pts, nullptr, 2));   // x = column 1, y = column 0
  CHECK_NUFFT(nufft_hip_execute(plan, d_c, d_f));
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(f.data(), d_f, f.size() * sizeof(float) * 2, hipMemcpyDeviceToHost));
  double num = 0, den = 0;
  for (int a2 = 0; a2 < N2; ++a2)
    for (int a1 = 0; a1 < N1; ++a1) {
      const int k1 = a1 - N1 / 2, k2 = a2 - N2 / 2;
      std::complex<double> s = 0;
      for (int64_t j = 0; j < M; ++j)
        s += std::complex<double>(c[j]) * std::exp(std::complex<double>(0, -(k1 * (double)pts[2 * j + 1] + k2 * (double)pts[2 * j])));
      num += std::norm(std::complex<double>(f[(size_t)a2 * N1 + a1]) - s);
      den += std::norm(s);
    }
  const double e1 = std::sqrt(num / den);
  nufft_hip_plan_destroy(plan);
  plan = nullptr;

  // type 2, backward, on the modes just computed: c'_j = sum_k f[k] exp(+i k.x_j)
  CHECK_NUFFT(nufft_hip_plan_create(&plan, NUFFT_HIP_TYPE_2, 2, dims, NUFFT_HIP_BACKWARD, 1, 1e-6, NUFFT_HIP_F32,
                                    &opts, nullptr, err, sizeof err));
  CHECK_NUFFT(nufft_hip_set_points(plan, M, d_pts + 1, d_pts, nullptr, 2));
  CHECK_NUFFT(nufft_hip_execute(plan, d_c, d_f));
  CHECK_HIP(hipDeviceSynchronize());
  std::vector<std::complex<float>> c2(M);
  CHECK_HIP(hipMemcpy(c2.data(), d_c, c2.size() * sizeof(float) * 2, hipMemcpyDeviceToHost));
  num = den = 0;
  for (int64_t j = 0; j < M; j += 7) {
    std::complex<double> s = 0;
    for (int a2 = 0; a2 < N2; ++a2)
      for (int a1 = 0; a1 < N1; ++a1)
        s += std::complex<double>(f[(size_t)a2 * N1 + a1]) *
             std::exp(std::complex<double>(0, (a1 - N1 / 2) * (double)pts[2 * j + 1] + (a2 - N2 / 2) * (double)pts[2 * j]));
    num += std::norm(std::complex<double>(c2[j]) - s);
    den += std::norm(s);
  }
  const double e2 = std::sqrt(num / den);
  nufft_hip_plan_destroy(plan);
  printf("abi_client: type-1 rel-l2 %.2e, type-2 rel-l2 %.2e (tol 1e-6)\n", e1, e2);
  (void)hipFree(d_pts); (void)hipFree(d_c); (void)hipFree(d_f);
  return (e1 < 1e-6 && e2 < 1e-6) ? 0 : 1;
}
