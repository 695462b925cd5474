// Micro-benchmark: the LDS read pattern of the thread-per-point interp kernels on gfx950. Every lane reads the 8 x 8
// cells (8 bytes each, row stride 72 cells) of ITS OWN stencil start inside a 71 x 72-cell tile, all lanes at the
// same (dy, dx) offset in the same instruction -- either as single ds_read_b64 (volatile LDS pointer: what
// lds_cell() in csrc/nufft_device.h does) or as the compiler writes the plain loop (pairs of neighbouring cells
// become ds_read2_b64). Start cells: "random" (points of a tile in arrival order), "cell order, 2.4 per cell"
// (config 3's density after a cell sort), "one cell" (every lane the same start: pure broadcast).
// Output: cycles per CELL read of a wavefront, per CU (two FMAs ride along), at 8 / 16 / 24 waves per CU (1-3 blocks of 512).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

constexpr int LS = 72, ROWS = 71, ITER = 400;
typedef float v2f __attribute__((ext_vector_type(2)));

template <bool SINGLE>
__global__ void bench(const int* __restrict__ starts, float* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float2* tile = reinterpret_cast<float2*>(smem);
  for (int i = threadIdx.x; i < LS * ROWS; i += blockDim.x) tile[i] = make_float2(1e-6f * i, 1.f);
  __syncthreads();
  const int cell0 = starts[blockIdx.x * blockDim.x + threadIdx.x];
  float re = 0.f, im = 0.f;
  for (int it = 0; it < ITER; ++it) {
    // (the start moves with the iteration, the same way for every lane: otherwise the plain loads are loop invariant)
    const int cell = (cell0 + 37 * it) & 4095;
    const float2* tp = tile + (cell >> 6) * LS + (cell & 63);
#pragma unroll
    for (int dy = 0; dy < 8; ++dy) {
#pragma unroll
      for (int dx = 0; dx < 8; ++dx) {
        float2 v;
        if constexpr (SINGLE) {
          typedef const volatile __attribute__((address_space(3))) v2f* lds_ptr;
          const v2f t = *(lds_ptr)(tp + dy * LS + dx);
          v = make_float2(t.x, t.y);
        } else {
          v = tp[dy * LS + dx];
        }
        re = fmaf(v.x, 1.0001f, re);
        im = fmaf(v.y, 0.9999f, im);
      }
    }
  }
  if (re == 123.f) out[blockIdx.x] = re + im;
}

template <bool SINGLE>
static double run(int cus, double clk, int nb, const int* dstarts, float* dout) {   // nb blocks of 8 waves per CU
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = (size_t)LS * ROWS * 8;
  bench<SINGLE><<<cus * nb, 512, lds>>>(dstarts, dout);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  bench<SINGLE><<<cus * nb, 512, lds>>>(dstarts, dout);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3 * clk / ((double)ITER * 64 * 8 * nb);   // cycles per cell read of a wave, per CU
}

int main() {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
  const int cus = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e3;
  float* dout;
  int* dstarts;
  const int maxthreads = cus * 24 * 64;
  if (hipMalloc(&dout, 1 << 20) != hipSuccess || hipMalloc(&dstarts, sizeof(int) * maxthreads) != hipSuccess) return 1;
  srand(7);
  for (int pat = 0; pat < 3; ++pat) {
    std::vector<int> st(maxthreads);
    for (int b = 0; b < maxthreads; b += 64) {      // one wavefront's 64 start cells
      std::vector<int> cell(64);
      if (pat == 0) { for (int& c : cell) c = rand() % 4096; }
      else if (pat == 1) { const int c0 = rand() % 4000; for (int l = 0; l < 64; ++l) cell[l] = std::min(4095, c0 + (int)(l / 2.4)); }
      else { const int c0 = rand() % 4096; for (int& c : cell) c = c0; }
      for (int l = 0; l < 64; ++l) st[b + l] = cell[l];
    }
    hipMemcpy(dstarts, st.data(), sizeof(int) * maxthreads, hipMemcpyHostToDevice);
    const char* names[3] = {"random starts", "cell order, 2.4 lanes per cell", "one cell"};
    for (int nb : {1, 2, 3})
      printf("%-31s %2d waves/CU: ds_read_b64 %.2f   compiler's pairing (ds_read2_b64) %.2f   [cycles per cell read of a wave, per CU]\n",
             names[pat], 8 * nb, run<true>(cus, clk, nb, dstarts, dout), run<false>(cus, clk, nb, dstarts, dout));
  }
  return 0;
}
