// Micro-benchmark: which lanes of a ds_add_u64 / ds_add_f64 wave-instruction conflict on LDS banks (gfx950)?
// Every lane adds at element (8 bytes) `pat[lane] + base`, base moving by the same amount for all lanes
// (bank relations between lanes stay fixed). Cycles per wave-instruction per CU at 16 waves per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

constexpr int ITER = 2000;
constexpr int LDSE = 8192;   // 8-byte elements of LDS per block (64 KB)

template <int MODE>
__global__ void bench(const int* __restrict__ pat, unsigned long long* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long lds[];
  const int tid = threadIdx.x;
  for (int i = tid; i < LDSE; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  const int p = pat[lane];
  int base = wave * 97;
  unsigned long long v = lane + 1;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int a = (base + u * 131 + p) & (LDSE - 1);
      if (MODE == 0) atomicAdd(&lds[a], v);
      else unsafeAtomicAdd(reinterpret_cast<double*>(&lds[a]), (double)v);
    }
    base = (base + 37) & (LDSE - 1);
  }
  __syncthreads();
  if (tid == 0) out[blockIdx.x] = lds[0] + v;
}

int main() {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
  const int cus = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e3;
  int* dpat; unsigned long long* dout;
  hipMalloc(&dpat, 64 * 4); hipMalloc(&dout, 1 << 16);
  struct P { std::string name; std::vector<int> pat; };
  std::vector<P> ps;
  auto mk = [&](const char* n, auto f) { P p; p.name = n; for (int l = 0; l < 64; ++l) p.pat.push_back(f(l)); ps.push_back(p); };
  mk("linear (lane l -> element l)", [](int l) { return l; });
  mk("l and l+32 distinct banks (element 2l: banks spread over 128)", [](int l) { return (l % 32) + 64 * (l / 32) + (l / 32) * 0; });
  mk("16-lane groups on 16 elements, groups 32 elements apart (2x32 grouping -> 2-way)", [](int l) { return (l % 16) + 32 * (l / 16); });
  mk("8-lane groups on 8 elements, 32 apart (16-lane grouping -> 2-way, 32-lane -> 4-way)", [](int l) { return (l % 8) + 32 * (l / 8); });
  mk("all lanes one bank, distinct addresses (64-way if one group)", [](int l) { return 32 * l; });
  mk("lanes l, l+32 same ADDRESS", [](int l) { return l % 32; });
  mk("8x8 patch, row stride 24 (spread_wave3_kernel today)", [](int l) { return (l & 7) + 24 * (l >> 3); });
  mk("8x8 patch, row stride 40 (2-D kernels)", [](int l) { return (l & 7) + 40 * (l >> 3); });
  // 6 x 3 x 3 cells, LS = 21, PS = 21 * 21 + 6 = 447 (== 31 mod 32): natural lane order
  mk("6x3x3 natural order, LS 21, PS 447", [](int l) { return l < 54 ? (l % 6) + 21 * ((l / 6) % 3) + 447 * (l / 18) : l; });
  mk("16-lane groups: lanes j, j+8 sixteen elements apart (2-way only if columns are mod 16)", [](int l) { const int j = l % 16; return (j % 8) + 16 * (j / 8) + 64 * (l / 16); });
  {   // the lane table of spread_dense3_kernel<6, 8>: 6 x 3 x 3 cells dealt to four 16-lane groups by column mod 16
    P p; p.name = "6x3x3 dealt to 4 groups of 16 by column (nufft_dense3.hip), LS 21, PS 441"; p.pat.assign(64, 0);
    bool used[4][16] = {}; int n[4] = {0, 0, 0, 0}, next[4] = {0, 16, 32, 48};
    for (int z = 0; z < 3; ++z) for (int y = 0; y < 3; ++y) for (int x = 0; x < 6; ++x) {
      const int cell = x + 21 * y + 441 * z, col = cell & 15; int best = -1;
      for (int g = 0; g < 4; ++g) if (!used[g][col] && (best < 0 || n[g] < n[best])) best = g;
      used[best][col] = true; ++n[best]; p.pat[next[best]++] = cell;
    }
    for (int g = 0; g < 4; ++g) { int col = 0; for (int l = next[g]; l < 16 * (g + 1); ++l) { while (used[g][col]) ++col; used[g][col] = true; p.pat[l] = col; } }
    ps.push_back(p);
  }
  for (auto& p : ps) {
    hipMemcpy(dpat, p.pat.data(), 64 * 4, hipMemcpyHostToDevice);
    float ms[2];
    for (int mode = 0; mode < 2; ++mode) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) bench<0><<<cus, 1024, LDSE * 8>>>(dpat, dout); else bench<1><<<cus, 1024, LDSE * 8>>>(dpat, dout);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      hipEventElapsedTime(&ms[mode], e0, e1);
    }
    const double ops = (double)ITER * 8 * 16;
    printf("%-90s ds_add_u64 %5.1f  ds_add_f64 %5.1f  cycles/wave-instr/CU\n", p.name.c_str(), ms[0] * 1e-3 * clk / ops, ms[1] * 1e-3 * clk / ops);
  }
  return 0;
}
