// Micro-benchmark for the r02 verdict's item 5: an atomic-free accumulate of the 2-D w = 8 spreader as a banded outer
// product on the matrix cores, G_block(16 x 16) += Ky^T (16 x 4) diag(c) Kx (4 x 16) with v_mfma_f32_16x16x4_f32, on
// synthetic cell-sorted points. What it measures: cycles per point and CU of the accumulate phase alone -- operands
// built from LDS-staged kernel values (8 per point and dimension, as spread_2d_w8_group_kernel stages them) with
// lane-indexed, window-masked reads, then one MFMA per (4 points, block, component) into wave-private accumulators --
// against the 13 cycles per point and CU of the LDS-atomic form (EXPERIMENTS.md section 4). Not measured: the kernel
// evaluation (same as today), the in-LDS sort by block, the end-of-tile reduction of the wave-private accumulators
// into the tile, and the periodic fp64 flush that fp32 accumulators would need on crowded tiles.
// Geometry: a wave works on points whose stencils start in one 16 x 16 block; an 8 x 8 stencil then meets 1, 2 or 4
// of the 2 x 2 blocks below and right of it (x spills when the start column is > 8: 7 / 16 of the points).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int NW = 12, NP = 64, ITER = 400;

struct Meta { int ox, oy; float cre, cim; };

// MODE 0: operands + MFMA (the real thing); 1: MFMA only (operands built once); 2: operands only (no MFMA)
template <int MODE>
__global__ __launch_bounds__(NW * 64) void bench(const float* __restrict__ kx_in, const float* __restrict__ ky_in,
                                                 const Meta* __restrict__ meta_in, float* __restrict__ out, int write_out) {
  __shared__ __attribute__((aligned(16))) float kxs[NW][NP * 8];
  __shared__ __attribute__((aligned(16))) float kys[NW][NP * 8];
  __shared__ __attribute__((aligned(16))) Meta metas[NW][NP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int src = (blockIdx.x * NW + wave) % 64;   // 64 different synthetic point sets
  for (int q = 0; q < 8; ++q) {
    kxs[wave][lane * 8 + q] = kx_in[(src * NP + lane) * 8 + q];
    kys[wave][lane * 8 + q] = ky_in[(src * NP + lane) * 8 + q];
  }
  metas[wave][lane] = meta_in[src * NP + lane];
  __syncthreads();
  v4f acc[2][2][2];   // [block row][block col][re / im]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int c = 0; c < 2; ++c) acc[a][b][c] = (v4f){0.f, 0.f, 0.f, 0.f};
  const int i = lane & 15, k = lane >> 4;
  float keep = 0.f;
  for (int it = 0; it < ITER; ++it) {
#pragma unroll 4
    for (int g = 0; g < NP / 4; ++g) {
      const int p = 4 * g + k;
      const Meta m = metas[wave][p];                       // ds_read_b128
      // which blocks does this group of 4 points meet? (wave-uniform)
      const bool sx = __any(m.ox > 8), sy = __any(m.oy > 8);
      float a[2], br[2], bi[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ia = i + 16 * h - m.oy, ib = i + 16 * h - m.ox;
        const float va = kys[wave][p * 8 + (ia & 7)], vb = kxs[wave][p * 8 + (ib & 7)];
        a[h] = (unsigned)ia < 8u ? va : 0.f;
        const float b = (unsigned)ib < 8u ? vb : 0.f;
        br[h] = b * m.cre;
        bi[h] = b * m.cim;
      }
      if (MODE == 2) { keep += a[0] + a[1] + br[0] + br[1] + bi[0] + bi[1]; continue; }
      if (MODE == 1) { a[0] = a[1] = 1.f + keep; br[0] = br[1] = bi[0] = bi[1] = 0.5f; }
#pragma unroll
      for (int hy = 0; hy < 2; ++hy) {
        if (hy == 1 && !sy) continue;
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) {
          if (hx == 1 && !sx) continue;
          acc[hy][hx][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[hy], br[hx], acc[hy][hx][0], 0, 0, 0);
          acc[hy][hx][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[hy], bi[hx], acc[hy][hx][1], 0, 0, 0);
        }
      }
    }
  }
  // D layout of v_mfma_f32_16x16x4_f32: lane l holds column l % 16, rows 4 (l / 16) + r, r = 0..3
  float s = keep;
#pragma unroll
  for (int hy = 0; hy < 2; ++hy)
#pragma unroll
    for (int hx = 0; hx < 2; ++hx)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (write_out && blockIdx.x == 0 && wave == 0)
            out[((c * 32 + 16 * hy + 4 * k + r) * 32) + 16 * hx + i] = acc[hy][hx][c][r];
          s += acc[hy][hx][c][r];
        }
  if (s == 1.2345f) out[0] = s;
}

int main() {
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, 0) != hipSuccess) return 1;
  const int cus = prop.multiProcessorCount;
  const double clk = prop.clockRate * 1e3;
  // 64 synthetic sets of 64 points, "cell-sorted": consecutive points share (nearly) the same start cell
  std::vector<float> kx(64 * NP * 8), ky(64 * NP * 8);
  std::vector<Meta> meta(64 * NP);
  srand(1);
  double blocks = 0;
  for (int s = 0; s < 64; ++s)
    for (int g = 0; g < NP / 4; ++g) {
      const int ox = rand() % 16, oy = rand() % 16;   // 2.4 points per cell ~ groups of 4 within one or two cells
      for (int k = 0; k < 4; ++k) {
        const int p = s * NP + 4 * g + k;
        meta[p] = {ox, oy, (float)rand() / RAND_MAX - .5f, (float)rand() / RAND_MAX - .5f};
        for (int q = 0; q < 8; ++q) { kx[p * 8 + q] = (float)rand() / RAND_MAX; ky[p * 8 + q] = (float)rand() / RAND_MAX; }
      }
      blocks += (ox > 8 ? 2 : 1) * (oy > 8 ? 2 : 1);
    }
  printf("blocks per 4-point group: %.2f (per point on average (1 + 7/16)^2 = 2.07)\n", blocks / (64 * NP / 4));
  float *dkx, *dky, *dout; Meta* dm;
  hipMalloc(&dkx, kx.size() * 4); hipMalloc(&dky, ky.size() * 4); hipMalloc(&dm, meta.size() * sizeof(Meta)); hipMalloc(&dout, 2 * 32 * 32 * 4);
  hipMemcpy(dkx, kx.data(), kx.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dky, ky.data(), ky.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dm, meta.data(), meta.size() * sizeof(Meta), hipMemcpyHostToDevice);
  // correctness of the operand build: wave 0 of block 0 (point set 0) against a direct sum
  hipMemset(dout, 0, 2 * 32 * 32 * 4);
  bench<0><<<1, NW * 64>>>(dkx, dky, dm, dout, 1);
  std::vector<float> got(2 * 32 * 32);
  hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost);
  double err = 0, nrm = 0;
  for (int c = 0; c < 2; ++c)
    for (int y = 0; y < 32; ++y)
      for (int x = 0; x < 32; ++x) {
        double ref = 0;
        for (int p = 0; p < NP; ++p) {
          const int dx = x - meta[p].ox, dy = y - meta[p].oy;
          if (dx >= 0 && dx < 8 && dy >= 0 && dy < 8) ref += (double)ky[p * 8 + dy] * kx[p * 8 + dx] * (c ? meta[p].cim : meta[p].cre);
        }
        ref *= ITER;
        err += (got[(c * 32 + y) * 32 + x] - ref) * (got[(c * 32 + y) * 32 + x] - ref); nrm += ref * ref;
      }
  printf("operand build check (wave 0 vs direct sum over its 64 points x %d passes): rel-l2 %.2e\n", ITER, sqrt(err / nrm));
  const char* names[3] = {"operands + MFMA", "MFMA only", "operands only"};
  for (int wgs = 1; wgs <= 2; ++wgs)
    for (int mode = 0; mode < 3; ++mode) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) bench<0><<<cus * wgs, NW * 64>>>(dkx, dky, dm, dout, 0);
        if (mode == 1) bench<1><<<cus * wgs, NW * 64>>>(dkx, dky, dm, dout, 0);
        if (mode == 2) bench<2><<<cus * wgs, NW * 64>>>(dkx, dky, dm, dout, 0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      const double pts_per_cu = (double)wgs * NW * NP * ITER;
      printf("%d workgroup(s) of %d waves per CU, %-16s: %6.2f cycles per point and CU  (LDS-atomic form today: 13)\n", wgs, NW,
             names[mode], ms * 1e-3 * clk / pts_per_cu);
    }
  return 0;
}
