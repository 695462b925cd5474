// Calibration of rocprofv3's FETCH_SIZE for GATHERS on gfx950 (r03 verdict: the x2 correction of
// MI355X_MICROARCH.md was calibrated on a wide streaming read; config 4's spread kernel gathers 8-byte strengths
// through the sort permutation, one 64-byte sector per point -- is its FETCH_SIZE to be doubled or not?).
// Kernels of KNOWN access counts over a table far larger than the 256 MiB Infinity Cache (so that every gathered
// sector comes from HBM):
//   stream16   reads N 16-byte elements in order                       (the guide's calibration case)
//   gather8    reads N  8-byte elements at uniformly random places      (the strength gather c[idx])
//   gather16   reads N 16-byte elements at uniformly random places      (a record gather through a permutation)
//   gather8x2  gather8 with two dependent passes over the same indices  (what spread_dense3_kernel did in r03)
// Run under `rocprofv3 --pmc FETCH_SIZE` (and TCC_EA0_RDREQ_sum / TCC_EA0_RDREQ_32B_sum in a second pass): the
// driver tools/fetch_calibration.sh divides the counter by N and prints bytes per gathered element next to the
// time-based lower bound (elements x sector / duration must stay below the ~6.3 TB/s the fabric delivers).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

__global__ void fill_idx(uint32_t* idx, int64_t n, uint64_t table) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint64_t x = (uint64_t)i * 0x9E3779B97F4A7C15ull + 0x7F4A7C15ull;   // splitmix64
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; x ^= x >> 31;
    idx[i] = (uint32_t)(x % table);
  }
}
__global__ void stream16(const uint4* __restrict__ t, int64_t n, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 v = t[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void gather8(const uint2* __restrict__ t, const uint32_t* __restrict__ idx, int64_t n, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint2 v = t[idx[i]];
    acc ^= v.x ^ v.y;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void gather16(const uint4* __restrict__ t, const uint32_t* __restrict__ idx, int64_t n, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint4 v = t[idx[i]];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void gather8x2(const uint2* __restrict__ t, const uint32_t* __restrict__ idx, int64_t n, uint32_t* __restrict__ sink) {
  // every workgroup makes two passes over ITS 4096 indices (the second one finds the lines gone from L2 when
  // 512 workgroups x 4096 x 64 B = 134 MB are in flight between them -- as in the r03 spread kernel)
  uint32_t acc = 0;
  for (int64_t base = (int64_t)blockIdx.x * 4096; base < n; base += (int64_t)gridDim.x * 4096)
    for (int pass = 0; pass < 2; ++pass)
      for (int k = threadIdx.x; k < 4096 && base + k < n; k += blockDim.x) {
        const uint2 v = t[idx[base + k]];
        acc ^= v.x ^ (v.y + pass);
      }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 100000000;          // gathered elements
  const uint64_t table_bytes = (uint64_t)(argc > 2 ? atoll(argv[2]) : 2048) << 20;   // MiB
  void* table; uint32_t* idx; uint32_t* sink;
  if (hipMalloc(&table, table_bytes) != hipSuccess || hipMalloc(&idx, n * 4) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
  hipMemset(table, 1, table_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  const int blocks = 2048, threads = 256;
  printf("N = %lld elements per kernel, table %llu MiB\n", (long long)n, (unsigned long long)(table_bytes >> 20));
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0); stream16<<<blocks, threads>>>((const uint4*)table, (int64_t)(table_bytes / 16), sink); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("stream16  : %8.3f ms  %7.1f GB/s of %llu bytes read in order\n", ms, table_bytes / ms * 1e-6, (unsigned long long)table_bytes);
    fill_idx<<<blocks, threads>>>(idx, n, table_bytes / 8);
    hipEventRecord(e0); gather8<<<blocks, threads>>>((const uint2*)table, idx, n, sink); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("gather8   : %8.3f ms  %7.2f ns per 1000 elements; 64-byte sectors at %7.1f GB/s, 128-byte lines at %7.1f GB/s\n", ms, ms * 1e9 / n, n * 64.0 / ms * 1e-6, n * 128.0 / ms * 1e-6);
    hipEventRecord(e0); gather8x2<<<512, 768>>>((const uint2*)table, idx, n, sink); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("gather8x2 : %8.3f ms  (two passes per 4096-index block)\n", ms);
    fill_idx<<<blocks, threads>>>(idx, n, table_bytes / 16);
    hipEventRecord(e0); gather16<<<blocks, threads>>>((const uint4*)table, idx, n, sink); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (rep) printf("gather16  : %8.3f ms  64-byte sectors at %7.1f GB/s\n", ms, n * 64.0 / ms * 1e-6);
  }
  return 0;
}
