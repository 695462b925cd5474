// Micro-benchmark: LDS READ cost by access width on gfx950, in the two patterns the
// spread kernels use: "linear" (lane i -> consecutive elements) and "bcast8"
// (8 distinct addresses per wave, each read by 8 lanes). Cycles per
// wave-instruction per CU at 16 waves/CU.
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int ITER = 2000;
constexpr int LDSW = 8192;   // words

template <int MODE, int PAT>
__global__ void bench(float* out) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < LDSW; i += blockDim.x) lds[i] = 1e-9f * i;
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  constexpr int EW = MODE == 0 ? 1 : MODE == 1 ? 2 : MODE == 2 ? 4 : 1;   // words per element
  const int el = PAT == 0 ? lane : (PAT == 1 ? (lane & 7) : (lane >> 3));
  int base = (wave * 512 + el * EW) & (LDSW - 1);
  float acc = 0.f;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int a = (base + u * 64 * EW) & (LDSW - 4);
      if (MODE == 0) acc += lds[a];
      else if (MODE == 1) { const float2 v = *reinterpret_cast<const float2*>(&lds[a & ~1]); acc += v.x + v.y; }
      else if (MODE == 2) { const float4 v = *reinterpret_cast<const float4*>(&lds[a & ~3]); acc += v.x + v.y + v.z + v.w; }
      else { acc += lds[a] + lds[(a + 512 + (lane >> 3) - (lane & 7)) & (LDSW - 1)]; }   // two b32 with different lane maps
    }
    base = (base + 8 * EW) & (LDSW - 1);
  }
  if (acc == 123.f) out[blockIdx.x] = acc;
}

template <int MODE, int PAT>
double run(int cus, double clk, int wpb, float* dout) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  bench<MODE, PAT><<<cus, wpb * 64, LDSW * 4>>>(dout);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  bench<MODE, PAT><<<cus, wpb * 64, LDSW * 4>>>(dout);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3 * clk / ((double)ITER * 8 * wpb);
}

int main() {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
  const int cus = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e3;
  float* dout;
  if (hipMalloc(&dout, 1 << 20) != hipSuccess) return 1;
  for (int wpb : {8, 16}) {
    printf("%2d waves/CU  linear: b32 %.1f b64 %.1f b128 %.1f 2xb32 %.1f | bcast(lane&7): b32 %.1f b64 %.1f b128 %.1f | bcast(lane>>3): b32 %.1f b64 %.1f b128 %.1f  [cycles/wave-instr/CU]\n",
           wpb, run<0, 0>(cus, clk, wpb, dout), run<1, 0>(cus, clk, wpb, dout), run<2, 0>(cus, clk, wpb, dout), run<3, 0>(cus, clk, wpb, dout),
           run<0, 1>(cus, clk, wpb, dout), run<1, 1>(cus, clk, wpb, dout), run<2, 1>(cus, clk, wpb, dout),
           run<0, 2>(cus, clk, wpb, dout), run<1, 2>(cus, clk, wpb, dout), run<2, 2>(cus, clk, wpb, dout));
  }
  return 0;
}
