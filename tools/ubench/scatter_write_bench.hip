// Micro-benchmark: what does a scattered write of S bytes cost on MI355X HBM?
// Models the tile sort's scatter (16-byte records written in short runs at random
// places of a 160 MB array). For segment sizes S = 16 .. 512 bytes, every segment
// written exactly once:
//   aligned   : segment start is a multiple of S (full 32/64/128-byte granules)
//   unaligned : segment start is a multiple of 16 bytes only (straddles granules)
// A segment is written by S/16 consecutive lanes with one 16-byte store each, in
// ONE wave-instruction ("together"), or by the same lane over S/16 consecutive loop
// iterations ("spread": what the sort kernel does, its records of one tile arrive over
// time). Reports effective GB/s of payload.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>

struct alignas(16) Rec { uint32_t a, b, c, d; };

// together: lane group writes one segment per instruction
__global__ void scatter_together(Rec* out, const uint32_t* seg_start, int64_t nseg, int lanes_per_seg) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t seg = t / lanes_per_seg;
  const int k = (int)(t - seg * lanes_per_seg);
  if (seg >= nseg) return;
  Rec r = {(uint32_t)t, 1u, 2u, 3u};
  out[(int64_t)seg_start[seg] + k] = r;
}

// spread: each thread owns segments and writes record k of all its segments in pass k
__global__ void scatter_spread(Rec* out, const uint32_t* seg_start, int64_t nseg, int recs_per_seg, int segs_per_thread) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int k = 0; k < recs_per_seg; ++k)
    for (int s = 0; s < segs_per_thread; ++s) {
      const int64_t seg = t + (int64_t)s * gridDim.x * blockDim.x;
      if (seg < nseg) {
        Rec r = {(uint32_t)t, (uint32_t)k, 2u, 3u};
        out[(int64_t)seg_start[seg] + k] = r;
      }
    }
}

int main() {
  const int64_t nrec = 10'000'000;   // 160 MB of records
  Rec* out;
  uint32_t* dstart;
  if (hipMalloc(&out, (nrec + 64) * sizeof(Rec)) != hipSuccess) return 1;
  if (hipMalloc(&dstart, nrec * sizeof(uint32_t)) != hipSuccess) return 1;
  // something to evict the caches between runs
  char* flush;
  const size_t flush_bytes = (size_t)1 << 30;
  hipMalloc(&flush, flush_bytes);
  std::mt19937_64 rng(1);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  printf("segment bytes | aligned together | unaligned together | aligned spread | unaligned spread   [GB/s of payload; 160 MB written per run]\n");
  for (int recs : {1, 2, 4, 5, 8, 16, 32}) {
    double res[4];
    for (int mode = 0; mode < 4; ++mode) {
      const bool aligned = (mode & 1) == 0, spread = mode >= 2;
      const int64_t nseg = nrec / recs;
      std::vector<uint32_t> start(nseg);
      // random permutation of the segment slots (disjoint, every record written once)
      for (int64_t i = 0; i < nseg; ++i) start[i] = (uint32_t)(i * recs);
      std::shuffle(start.begin(), start.end(), rng);
      if (!aligned && recs > 1)
        for (auto& v : start) v += (uint32_t)(recs / 2);   // half a segment off: every segment straddles an S-aligned boundary
      hipMemcpy(dstart, start.data(), nseg * sizeof(uint32_t), hipMemcpyHostToDevice);
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipMemsetAsync(flush, rep, flush_bytes);
        hipEventRecord(e0);
        if (!spread) {
          const int64_t threads = nseg * recs;
          scatter_together<<<(unsigned)((threads + 255) / 256), 256>>>(out, dstart, nseg, recs);
        } else {
          const int blocks = 512 * 4, tpb = 256;
          const int spt = (int)((nseg + (int64_t)blocks * tpb - 1) / ((int64_t)blocks * tpb));
          scatter_spread<<<blocks, tpb>>>(out, dstart, nseg, recs, spt);
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = std::min(best, ms);
      }
      res[mode] = (double)nrec * 16 / (best * 1e-3) / 1e9;
    }
    printf("%5d B | %8.0f | %8.0f | %8.0f | %8.0f\n", recs * 16, res[0], res[1], res[2], res[3]);
  }
  return 0;
}
