// What bounds the tile + halo write-out of the 3-D spread kernels? (r04: 8192-65536 tiles of 23 x 23 x 15 cells, two
// global_atomic_add_f32 per cell, run at ~1.1 TB/s = 2.7e11 atomic operations per second whatever the point count.)
// Every workgroup adds `rows` rows of 23 cells (a 64-bit cell = (re, im)) of a tile + halo block into a 1 GiB grid
// of 512^3 cells at its tile's place, neighbouring tiles overlapping in their halos as in the product:
//   f32x2   two float atomics per cell: consecutive lanes carry (re, im) of consecutive cells   (the product's write-out)
//   u64     ONE 64-bit integer atomic per cell                                                    (packed fixed-point grid)
//   f64     one double atomic per cell (same bytes as u64; is the float unit slower?)
//   f32     one float atomic per cell (half the bytes of f32x2: is the rate per operation or per byte?)
//   store64 plain 8-byte stores, no atomics (the floor of the access pattern itself)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

constexpr int NF = 512, T0 = 16, T1 = 16, T2 = 8, W = 8, L0 = T0 + W - 1, L1 = T1 + W - 1, L2 = T2 + W - 1;

template <int MODE>
__global__ __launch_bounds__(768) void writeout(unsigned long long* grid, int ntile0, int ntile1) {
  const int tb = blockIdx.x;
  const int t0 = tb % ntile0, t1 = (tb / ntile0) % ntile1, t2 = tb / (ntile0 * ntile1);
  const int o0 = t0 * T0, o1 = t1 * T1, o2 = t2 * T2;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int row = wave; row < L1 * L2; row += 12) {
    const int a1 = row % L1, a2 = row / L1;
    const int g1 = (o1 + a1) % NF, g2 = (o2 + a2) % NF;
    const int64_t rowbase = (int64_t)NF * (g1 + (int64_t)NF * g2);
    if (MODE == 0) {          // f32x2
      float* out = reinterpret_cast<float*>(grid);
      for (int e = lane; e < 2 * L0; e += 64) {
        const int a0 = e >> 1, comp = e & 1;
        unsafeAtomicAdd(&out[2 * (rowbase + (o0 + a0) % NF) + comp], 1.0f + comp);
      }
    } else if (MODE == 1) {   // u64
      if (lane < L0) atomicAdd(&grid[rowbase + (o0 + lane) % NF], 0x0000000100000001ull);
    } else if (MODE == 2) {   // f64
      double* out = reinterpret_cast<double*>(grid);
      if (lane < L0) unsafeAtomicAdd(&out[rowbase + (o0 + lane) % NF], 1.0);
    } else if (MODE == 3) {   // f32, one per cell
      float* out = reinterpret_cast<float*>(grid);
      if (lane < L0) unsafeAtomicAdd(&out[2 * (rowbase + (o0 + lane) % NF)], 1.0f);
    } else {                  // plain stores
      if (lane < L0) grid[rowbase + (o0 + lane) % NF] = (unsigned long long)tb;
    }
  }
}

// colour c = parity of the tile coordinates: tiles of one colour have disjoint tile + halo blocks (halo 7 < tile 8..16),
// so a launch over one colour may read, add and store without atomics; 8 launches cover the grid
__global__ __launch_bounds__(768) void writeout_rmw(float2* grid, int ntile0, int ntile1, int colour, int first_touch) {
  const int h0 = ntile0 / 2, h1 = ntile1 / 2;
  const int tb = blockIdx.x;
  const int t0 = 2 * (tb % h0) + (colour & 1), t1 = 2 * ((tb / h0) % h1) + ((colour >> 1) & 1), t2 = 2 * (tb / (h0 * h1)) + (colour >> 2);
  const int o0 = t0 * T0, o1 = t1 * T1, o2 = t2 * T2;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int row = wave; row < L1 * L2; row += 12) {
    const int a1 = row % L1, a2 = row / L1;
    const int g1 = (o1 + a1) % NF, g2 = (o2 + a2) % NF;
    const int64_t rowbase = (int64_t)NF * (g1 + (int64_t)NF * g2);
    if (lane < L0) {
      float2* p = &grid[rowbase + (o0 + lane) % NF];
      float2 v = first_touch ? make_float2(0.f, 0.f) : *p;
      v.x += 1.0f; v.y += 2.0f;
      *p = v;
    }
  }
}

int main() {
  unsigned long long* grid;
  const size_t bytes = (size_t)NF * NF * NF * 8;
  if (hipMalloc(&grid, bytes) != hipSuccess) return 1;
  hipMemset(grid, 0, bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int n0 = NF / T0, n1 = NF / T1, n2 = NF / T2, ntiles = n0 * n1 * n2;
  const double cells = (double)ntiles * L0 * L1 * L2;
  printf("%d tiles of %d x %d x %d cells (+ halo: %d x %d x %d), %.3g cells written per launch, 12 waves per workgroup\n", ntiles, T0, T1, T2, L0, L1, L2, cells);
  const char* names[5] = {"f32x2 (product)", "u64", "f64", "f32 (one per cell)", "store64"};
  for (int mode = 0; mode < 5; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      switch (mode) {
        case 0: writeout<0><<<ntiles, 768>>>(grid, n0, n1); break;
        case 1: writeout<1><<<ntiles, 768>>>(grid, n0, n1); break;
        case 2: writeout<2><<<ntiles, 768>>>(grid, n0, n1); break;
        case 3: writeout<3><<<ntiles, 768>>>(grid, n0, n1); break;
        default: writeout<4><<<ntiles, 768>>>(grid, n0, n1); break;
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double ops = cells * (mode == 0 ? 2 : 1), by = cells * (mode == 3 ? 4 : 8);
    printf("%-20s %8.3f ms  %7.1f G atomic operations/s  %7.1f GB/s of cells\n", names[mode], best, ops / best * 1e-6, by / best * 1e-6);
  }
  for (int variant = 0; variant < 2; ++variant) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int c = 0; c < 8; ++c) writeout_rmw<<<ntiles / 8, 768>>>(reinterpret_cast<float2*>(grid), n0, n1, c, variant == 1 && c == 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("%-20s %8.3f ms  (8 colour launches, load + add + store without atomics%s)  %7.1f GB/s of cells\n", "rmw by colour", best,
           variant ? "; colour 0 stores only" : "", cells * 8 / best * 1e-6);
  }
  return 0;
}
