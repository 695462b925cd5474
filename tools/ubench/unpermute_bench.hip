// r06 (r05 verdict, item 5a): what would UN-PERMUTING the type-2 results cost?
//
// interp_point_kernel writes c[idx] in the caller's point order: one 8-byte store per point, each its own write
// transaction -- 35 % of the kernel at config 3 (EXPERIMENTS.md 11.14; storing in sorted order instead: 231 -> 152 us).
// The alternative: the kernel stores csorted[j] (coalesced) and a second kernel restores the caller's order. This
// program times that second kernel in its IDEAL form, on the permutation config 3 has (M = 1e7 uniform points, 1024
// tiles, the staged scatter's chunks of 8192 consecutive input points, inside a chunk the points of a tile get
// consecutive sorted positions -- runs of ~8): one workgroup per chunk reads the chunk's (sorted position, local
// index) pairs in tile order (what the staged scatter could write as it goes: 6 bytes per point), gathers
// csorted[P] with consecutive lanes on the 64-byte runs, parks the values in LDS at their local index and writes the
// chunk of c coalesced. Also timed: the naive form (thread per point, c[i] = csorted[pos[i]]) and the plain copy
// of 80 MB (the floor). Kill line of the experiment: interp (sorted store) + un-permute must beat the scattered
// store by 3 % of the step: the second kernel has < 62 us (231 - 152 - 0.03 x 380 us, less the pairs the scatter
// would have to write).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <numeric>
#include <random>
#include <vector>

constexpr int kChunk = 8192;

__global__ __launch_bounds__(1024) void unpermute_chunks(const float2* __restrict__ csorted, const uint32_t* __restrict__ pos,
                                                         const uint16_t* __restrict__ loc, float2* __restrict__ c, int64_t M) {
  __shared__ float2 buf[kChunk];
  const int64_t base = (int64_t)blockIdx.x * kChunk;
  const int n = (int)std::min<int64_t>(kChunk, M - base);
  for (int r = threadIdx.x; r < n; r += blockDim.x) buf[loc[base + r]] = csorted[pos[base + r]];
  __syncthreads();
  for (int r = threadIdx.x; r < n; r += blockDim.x) c[base + r] = buf[r];
}
__global__ void unpermute_naive(const float2* __restrict__ csorted, const uint32_t* __restrict__ pos_of, float2* __restrict__ c, int64_t M) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M) c[i] = csorted[pos_of[i]];
}
__global__ void copy_kernel(const float2* __restrict__ a, float2* __restrict__ b, int64_t M) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < M) b[i] = a[i];
}
// what the interp kernel does today: scattered 8-byte stores in the caller's order, from the sorted order
__global__ void scatter_store(const float2* __restrict__ csorted, const uint32_t* __restrict__ idx, float2* __restrict__ c, int64_t M) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < M) c[idx[j]] = csorted[j];
}

int main() {
  const int64_t M = 10'000'000;
  const int ntiles = 1024;
  std::mt19937_64 rng(3);
  std::vector<uint32_t> tile(M);
  for (auto& t : tile) t = (uint32_t)(rng() % ntiles);
  // stable counting sort by tile = the sort's order (input order inside a tile)
  std::vector<uint32_t> start(ntiles + 1, 0), pos_of(M), idx(M);
  for (int64_t i = 0; i < M; ++i) ++start[tile[i] + 1];
  for (int t = 0; t < ntiles; ++t) start[t + 1] += start[t];
  {
    std::vector<uint32_t> cur(start.begin(), start.end() - 1);
    for (int64_t i = 0; i < M; ++i) { pos_of[i] = cur[tile[i]]++; idx[pos_of[i]] = (uint32_t)i; }
  }
  // per chunk, tile order: (sorted position, local index)
  std::vector<uint32_t> pos(M);
  std::vector<uint16_t> loc(M);
  for (int64_t b = 0; b < M; b += kChunk) {
    const int n = (int)std::min<int64_t>(kChunk, M - b);
    std::vector<int> ord(n);
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int x, int y) { return tile[b + x] < tile[b + y]; });
    for (int r = 0; r < n; ++r) { pos[b + r] = pos_of[b + ord[r]]; loc[b + r] = (uint16_t)ord[r]; }
  }
  float2 *csorted, *c;
  uint32_t *dpos, *dposof, *didx;
  uint16_t* dloc;
  hipMalloc(&csorted, M * 8); hipMalloc(&c, M * 8); hipMalloc(&dpos, M * 4); hipMalloc(&dposof, M * 4); hipMalloc(&didx, M * 4); hipMalloc(&dloc, M * 2);
  hipMemset(csorted, 1, M * 8);
  hipMemcpy(dpos, pos.data(), M * 4, hipMemcpyHostToDevice);
  hipMemcpy(dposof, pos_of.data(), M * 4, hipMemcpyHostToDevice);
  hipMemcpy(didx, idx.data(), M * 4, hipMemcpyHostToDevice);
  hipMemcpy(dloc, loc.data(), M * 2, hipMemcpyHostToDevice);
  char* flush;
  const size_t flush_bytes = (size_t)1 << 30;
  hipMalloc(&flush, flush_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const unsigned nchunk = (unsigned)((M + kChunk - 1) / kChunk), nb = (unsigned)((M + 255) / 256);
  for (int which = 0; which < 4; ++which) {
    float best = 1e9f, sum = 0.f;
    for (int rep = 0; rep < 6; ++rep) {
      hipMemsetAsync(flush, rep, flush_bytes, 0);   // evict (the Infinity Cache holds 256 MB)
      hipEventRecord(e0, 0);
      if (which == 0) copy_kernel<<<nb, 256>>>(csorted, c, M);
      else if (which == 1) unpermute_chunks<<<nchunk, 1024>>>(csorted, dpos, dloc, c, M);
      else if (which == 2) unpermute_naive<<<nb, 256>>>(csorted, dposof, c, M);
      else scatter_store<<<nb, 256>>>(csorted, didx, c, M);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) { best = std::min(best, ms); sum += ms; }
    }
    const char* names[] = {"copy 80 MB -> 80 MB (floor)", "un-permute, one workgroup per chunk of 8192 (runs of ~8 gathered, LDS, coalesced store)",
                           "un-permute, naive gather c[i] = csorted[pos[i]]", "scattered store c[idx[j]] = csorted[j] (what the interp kernel's store amounts to)"};
    printf("%-95s best %7.1f us  mean %7.1f us\n", names[which], best * 1e3f, sum / 5 * 1e3f);
  }
  return 0;
}
