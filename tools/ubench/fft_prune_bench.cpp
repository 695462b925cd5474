// Micro-benchmark: full 2-D in-place C2C (what the plan uses) against a pruned pair
// of 1-D passes (all rows along x, then only the N/2 + N/2 columns the deconvolve reads).
#include <hip/hip_runtime.h>
#include <rocfft/rocfft.h>
#include <cstdio>
#include <vector>
#define CK(x) do { if ((x) != 0) { printf("fail line %d\n", __LINE__); return 1; } } while (0)

template <typename F> static float time_it(hipStream_t s, int reps, F&& fn) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) fn();
  hipEventRecord(a, s);
  for (int i = 0; i < reps; ++i) fn();
  hipEventRecord(b, s); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
  const size_t nf = argc > 1 ? atoi(argv[1]) : 2048, N = nf / 2;
  rocfft_setup();
  void* buf; CK(hipMalloc(&buf, nf * nf * 8)); CK(hipMemset(buf, 0, nf * nf * 8));
  hipStream_t s = nullptr;
  auto mk = [&](rocfft_plan* p, size_t dims, const size_t* len, size_t batch, rocfft_plan_description d) {
    return rocfft_plan_create(p, rocfft_placement_inplace, rocfft_transform_type_complex_forward, rocfft_precision_single, dims, len, batch, d);
  };
  // full 2-D
  rocfft_plan p2; size_t l2[2] = {nf, nf}; CK(mk(&p2, 2, l2, 1, nullptr));
  // rows along x
  rocfft_plan px; size_t l1[1] = {nf}; CK(mk(&px, 1, l1, nf, nullptr));
  // columns along y: stride nf, distance 1, batch N/2 (launched twice, two column blocks)
  rocfft_plan_description d; CK(rocfft_plan_description_create(&d));
  size_t str[1] = {nf};
  CK(rocfft_plan_description_set_data_layout(d, rocfft_array_type_complex_interleaved, rocfft_array_type_complex_interleaved, nullptr, nullptr, 1, str, 1, 1, str, 1));
  rocfft_plan py; CK(mk(&py, 1, l1, N / 2, d));
  size_t wb = 0, w; rocfft_plan_get_work_buffer_size(p2, &w); wb = w > wb ? w : wb;
  rocfft_plan_get_work_buffer_size(px, &w); wb = w > wb ? w : wb;
  rocfft_plan_get_work_buffer_size(py, &w); wb = w > wb ? w : wb;
  void* work = nullptr; if (wb) CK(hipMalloc(&work, wb));
  rocfft_execution_info info; CK(rocfft_execution_info_create(&info));
  if (wb) CK(rocfft_execution_info_set_work_buffer(info, work, wb));
  void* b0[1] = {buf};
  void* bA[1] = {buf};
  void* bB[1] = {(char*)buf + (nf - N / 2) * 8};
  const float t_full = time_it(s, 50, [&] { rocfft_execute(p2, b0, nullptr, info); });
  const float t_x = time_it(s, 50, [&] { rocfft_execute(px, b0, nullptr, info); });
  const float t_y = time_it(s, 50, [&] { rocfft_execute(py, bA, nullptr, info); rocfft_execute(py, bB, nullptr, info); });
  printf("nf=%zu: full 2-D %.1f us | rows(x) %.1f us + pruned columns(y, 2 x %zu) %.1f us = %.1f us (work buffer %zu B)\n",
         nf, t_full, t_x, N / 2, t_y, t_x + t_y, wb);
  return 0;
}
