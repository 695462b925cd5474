// Issue rate of the vector instructions in the inner loop of the 3-D fixed-point spreaders (gfx950): cycles per
// wave-instruction and SIMD with W waves per SIMD, each wave running a chain-free stream of ONE instruction kind
// (16 independent destinations), and of the mix one plane of spread_patch3_kernel issues with / without its ds_add_u64.
//   make -C tools/ubench valu_rate_bench && tools/ubench/valu_rate_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

enum { OP_FMA, OP_PK_FMA, OP_PK_MUL, OP_CVT_RPI, OP_CVT_I32, OP_RNDNE, OP_READLANE, OP_MAX, OP_PLANE, OP_PLANE_LDS, OP_LDS_ONLY, OP_PLANE_FMAMAGIC, OP_COUNT };
const char* kNames[] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_cvt_rpi_i32_f32", "v_cvt_i32_f32", "v_rndne_f32", "v_readlane_b32",
                        "v_max_f32", "plane mix (readlane + pk_mul + 2 cvt_rpi), no LDS", "plane mix + ds_add_u64", "ds_add_u64 alone",
                        "plane mix with pk_fma magic conversion + ds_add_u64"};

typedef float v2f __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(unsigned long long* out, int iters, float seed) {
  __shared__ unsigned long long cell[4 * 64 * 9];
  for (int i = threadIdx.x; i < 4 * 64 * 9; i += 256) cell[i] = 0ull;
  __syncthreads();
  float a[16];
  v2f p[8];
#pragma unroll
  for (int k = 0; k < 16; ++k) a[k] = seed + k + threadIdx.x * 1e-3f;
#pragma unroll
  for (int k = 0; k < 8; ++k) p[k] = (v2f){seed + k, seed - k};
  const float b = seed * 0.999f, c = seed * 1e-3f;
  const v2f pb = {b, b}, pc = {c, c};
  // conflict-free 8-byte cells: lane's own column (ds_add_u64 at 7.0 cycles per wave-instruction)
  unsigned long long* mine = cell + (threadIdx.x >> 6) * 64 * 9 + (threadIdx.x & 63);
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (OP == OP_FMA) {
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));
    } else if constexpr (OP == OP_PK_FMA) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k]) : "v"(pb), "v"(pc));
    } else if constexpr (OP == OP_PK_MUL) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k]) : "v"(pb));
    } else if constexpr (OP == OP_CVT_RPI) {
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("v_cvt_rpi_i32_f32 %0, %0" : "+v"(a[k]));
    } else if constexpr (OP == OP_CVT_I32) {
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[k]));
    } else if constexpr (OP == OP_RNDNE) {
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[k]));
    } else if constexpr (OP == OP_READLANE) {
      int s;
#pragma unroll
      for (int k = 0; k < 16; ++k) { asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(s) : "v"(a[k])); asm volatile("" :: "s"(s)); }
    } else if constexpr (OP == OP_MAX) {
#pragma unroll
      for (int k = 0; k < 16; ++k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));
    } else if constexpr (OP == OP_PLANE || OP == OP_PLANE_LDS || OP == OP_PLANE_FMAMAGIC) {
      // 8 planes of one point, as the kernel issues them
#pragma unroll
      for (int dz = 0; dz < 8; ++dz) {
        int s;
        asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(s) : "v"(a[dz]));
        const float kz = __builtin_bit_cast(float, s);
        v2f v = p[0] * (v2f){kz, kz};
        unsigned long long word;
        if constexpr (OP == OP_PLANE_FMAMAGIC) {
          const v2f m = __builtin_elementwise_fma(p[0], (v2f){kz, kz}, (v2f){12582912.f, 12582912.f});
          word = __builtin_bit_cast(unsigned long long, m) + 0x1234ull;
        } else {
          int r0, r1;
          asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r0) : "v"(v.x));
          asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r1) : "v"(v.y));
          word = ((unsigned long long)(unsigned)r1 << 32) | (unsigned)r0;
        }
        if constexpr (OP == OP_PLANE) asm volatile("" :: "v"(word));
        else __hip_atomic_fetch_add(mine + dz * 64, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    } else if constexpr (OP == OP_LDS_ONLY) {
#pragma unroll
      for (int dz = 0; dz < 8; ++dz) __hip_atomic_fetch_add(mine + dz * 64, 0x100000001ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  float acc = 0.f;
#pragma unroll
  for (int k = 0; k < 16; ++k) acc += a[k];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc += p[k].x + p[k].y;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (acc == 1.2345f || cell[threadIdx.x] == 0x77ull) out[0] = 0;
}

template <int OP>
int run(unsigned long long* d, int W, int iters) {
  const int blocks = 256 * W;   // 256 CUs x W workgroups of 4 waves: W waves per SIMD
  std::vector<unsigned long long> h(blocks);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float ms = 0.f;
  for (int rep = 0; rep < 2; ++rep) {
    CHECK(hipEventRecord(e0));
    rate_kernel<OP><<<blocks, 256>>>(d, iters, 1.0f + rep);
    CHECK(hipEventRecord(e1));
    CHECK(hipMemcpy(h.data(), d, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
  }
  std::sort(h.begin(), h.end());
  const double cyc = (double)h[blocks / 2];
  const int per_iter = (OP >= OP_PLANE) ? 8 : 16;
  printf("  %-58s W=%d: %6.2f cycles per %s and SIMD   (kernel %.3f ms = %.2f ns per item and SIMD; counter %.0f MHz)\n", kNames[OP], W,
         cyc / ((double)iters * per_iter * W), OP >= OP_PLANE ? "plane (wave)" : "wave-instruction", ms,
         ms * 1e6 / ((double)iters * per_iter * W), cyc / (ms * 1e3));
  return 0;
}

int main() {
  unsigned long long* d;
  CHECK(hipMalloc(&d, sizeof(unsigned long long) * 256 * 8));
  const int iters = 2000;
  for (int W : {1, 2, 4, 6, 8}) {
    if (run<OP_FMA>(d, W, iters) || run<OP_PK_FMA>(d, W, iters) || run<OP_PK_MUL>(d, W, iters) || run<OP_CVT_RPI>(d, W, iters) ||
        run<OP_CVT_I32>(d, W, iters) || run<OP_RNDNE>(d, W, iters) || run<OP_READLANE>(d, W, iters) || run<OP_MAX>(d, W, iters) ||
        run<OP_PLANE>(d, W, iters) || run<OP_PLANE_LDS>(d, W, iters) || run<OP_LDS_ONLY>(d, W, iters) || run<OP_PLANE_FMAMAGIC>(d, W, iters))
      return 1;
    printf("\n");
  }
  return 0;
}
