// Micro-benchmark: LDS accumulate throughput on gfx950 for the forms the
// spread kernel could use. Each wave issues ITER x UNROLL ops on conflict-free
// addresses (lane i -> word base + i, base rotates), per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITER = 2000;
constexpr int LDSW = 8192;   // words of LDS per block

template <int MODE>
__global__ void bench(float* out, int stride_words) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < LDSW; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  const int lane = tid & 63;
  const int wave = tid >> 6;
  float v = 1.0f + lane * 1e-3f;
  int base = (wave * 512 + lane * stride_words) & (LDSW - 1);
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int a = (base + u * 64 * 4 / 4) & (LDSW - 1);
      if (MODE == 0) unsafeAtomicAdd(&lds[a], v);                                   // ds_add_f32
      else if (MODE == 1) v += unsafeAtomicAdd(&lds[a], v) * 1e-9f;                 // ds_add_rtn_f32
      else if (MODE == 2) atomicAdd(reinterpret_cast<unsigned*>(&lds[a]), (unsigned)lane);  // ds_add_u32
      else if (MODE == 3) atomicAdd(reinterpret_cast<unsigned long long*>(&lds[a & ~1]), (unsigned long long)lane); // ds_add_u64
      else if (MODE == 4) unsafeAtomicAdd(reinterpret_cast<double*>(&lds[a & ~1]), (double)v);   // ds_add_f64
      else if (MODE == 5) { float t = lds[a]; lds[a] = t + v; }                     // plain RMW (dependent)
      else if (MODE == 6) { lds[a] = v; }                                            // plain write
      else if (MODE == 7) { v += lds[a]; }                                           // plain read
      else if (MODE == 8) atomicMax(reinterpret_cast<int*>(&lds[a]), lane);          // ds_max_i32
      // ds_add_f64 with part of the wave active: does the cost scale with the active lanes?
      else if (MODE == 9) { if (lane < 32) unsafeAtomicAdd(reinterpret_cast<double*>(&lds[a & ~1]), (double)v); }
      else if (MODE == 10) { if (lane < 16) unsafeAtomicAdd(reinterpret_cast<double*>(&lds[a & ~1]), (double)v); }
      else if (MODE == 11) { if (lane < 8) unsafeAtomicAdd(reinterpret_cast<double*>(&lds[a & ~1]), (double)v); }
      else if (MODE == 12) { if ((lane & 7) == 0) unsafeAtomicAdd(reinterpret_cast<double*>(&lds[a & ~1]), (double)v); }
    }
    base = (base + 37) & (LDSW - 1);
  }
  __syncthreads();
  if (tid == 0) out[blockIdx.x] = lds[0] + v;
}

template <int MODE>
float run(int blocks, int threads, int stride_words, float* dout) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  bench<MODE><<<blocks, threads, LDSW * 4>>>(dout, stride_words);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  bench<MODE><<<blocks, threads, LDSW * 4>>>(dout, stride_words);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e3;
  printf("device %s CUs %d clock %.0f MHz\n", prop.name, cus, clk / 1e6);
  float* dout;
  CHECK(hipMalloc(&dout, 1 << 20));
  const char* names[] = {"ds_add_f32", "ds_add_rtn_f32", "ds_add_u32", "ds_add_u64", "ds_add_f64", "rmw_plain", "write_b32", "read_b32", "ds_max_i32",
                         "f64_lanes<32", "f64_lanes<16", "f64_lanes<8", "f64_lanes%8==0"};
  for (int stride : {1, 2}) {
    for (int wpb : {1, 4, 8, 16}) {
      const int threads = wpb * 64;
      const int blocks = cus;  // one block per CU
      printf("stride %d words, %2d waves/CU:", stride, wpb);
      float ms[13];
      ms[0] = run<0>(blocks, threads, stride, dout); ms[1] = run<1>(blocks, threads, stride, dout);
      ms[2] = run<2>(blocks, threads, stride, dout); ms[3] = run<3>(blocks, threads, stride, dout);
      ms[4] = run<4>(blocks, threads, stride, dout); ms[5] = run<5>(blocks, threads, stride, dout);
      ms[6] = run<6>(blocks, threads, stride, dout); ms[7] = run<7>(blocks, threads, stride, dout);
      ms[8] = run<8>(blocks, threads, stride, dout);
      ms[9] = run<9>(blocks, threads, stride, dout); ms[10] = run<10>(blocks, threads, stride, dout);
      ms[11] = run<11>(blocks, threads, stride, dout); ms[12] = run<12>(blocks, threads, stride, dout);
      for (int m = 0; m < 13; ++m) {
        const double ops = (double)ITER * 8 * wpb;            // wave-instructions per CU
        const double cyc = ms[m] * 1e-3 * clk / ops;          // cycles per wave-instruction per CU
        printf("  %s %.1f", names[m], cyc);
      }
      printf("  [cycles/wave-instr/CU]\n");
    }
  }
  return 0;
}
