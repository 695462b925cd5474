// Internal declarations shared by the host plan (nufft_plan.cpp), the op-level
// host logic (nufft_op.cpp) and the gfx950 kernels (nufft_kernels.hip).
#ifndef NUFFT_HIP_INTERNAL_H_
#define NUFFT_HIP_INTERNAL_H_

#include "nufft_experiment.h"   // (first: refuses experiment macros without -DNUFFT_EXPERIMENT_BUILD)

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "nufft_hip.h"

namespace nufft_hip {

constexpr int kMaxW = 16;        // kMaxKernelWidth, reference nufft_plan.h:68
constexpr int kMaxCoef = 24;     // polynomial terms per stencil cell
constexpr int kWaveCoef = 12;    // fixed term count of the wave-per-point kernels
constexpr int kBlock = 256;      // threads per workgroup of the tile kernels

// Geometry of one plan, passed by value to every kernel.
struct Geom {
  int rank;
  int w;            // kernel width
  int ncoef;        // polynomial terms
  int nf[3];        // fine grid, x fastest
  int tile[3];      // tile size in fine cells
  int tile_shift[3];  // log2(tile) when tile is a power of two, else -1
  int fold_pow2;    // every used dimension's tile edge is a power of two (the sort kernels' short fold path; 0 with tuning QFOLD_OFF)
  int ntile[3];     // tiles per dimension
  int ldim[3];      // LDS tile extent = tile + w - 1
  int lstride;      // padded LDS row length (>= ldim[0])
  int ntiles;       // product of ntile
  int max_sub;      // points per subproblem
  int nmodes[3];    // N, x fastest
  int fixed_point;  // 3-D float spread accumulates packed 32+32-bit fixed point in LDS
  int split_reim;   // 3-D float fp64-plane spread: real and imaginary parts in separate launches
  int fx_max_subs;  // fixed_point: tiles with more subproblems than this are left to the fp64-plane kernels
                    // (launched behind the fixed-point one; each kind exits on the other's tiles)
  int cell_sorted;  // records of each subproblem are ordered by stencil start cell (set per set_points)
  int nitems;       // point sets sorted together (batched per-item points); tiles are then composite:
  int ntiles_item;  //   item * ntiles_item + tile, ntiles = nitems * ntiles_item
  int sparse_auto;  // spread_method AUTO: launch_spread may pick the LDS-free kernel for sparse point sets
  int fused;        // float records carry the strength: 2-D FusedRec (instead of the point index), 3-D FusedRec3 (32 bytes)
  int line;         // 1-D plan whose interpolation runs on interp_line_kernel (nufft_line.hip; spread_method AUTO)
  int wide;         // w = 9..16, rank 2 / 3: tiles and LDS strides of the 16 x 4-lane kernels (nufft_wide.hip)
  int sub_small;    // > 0: scan_tiles_kernel caps the subproblems at this many points instead of max_sub when the
                    // point set turns out clustered (a tile holds more than 1.5x the average): 2-D type-2 plans
  float fx_headroom;  // fixed-point accumulation: bound on prod_d max|P(z)| of the fitted kernel (>= 1)
  int tuning;       // nufft_hip_options.tuning (NUFFT_HIP_TUNE_* bits; read by the host-side launchers only)
  // Tile numbering: tiles are numbered super-tile by super-tile (2^sup_shift[d] tiles per dimension each, nsup[d]
  // super-tiles per dimension; id = super-tile << sum(sup_shift) | tile inside it, x fastest in both parts), which
  // is what the two-level sort of large 3-D tile sets sorts by. sup_shift = {0, 0, 0}, nsup = ntile: the plain
  // x-fastest numbering.
  int sup_shift[3];
  int nsup[3];
  // 3-D float fixed-point plans at w = 7, 8 (spread_patch3_kernel, nufft_dense3.hip): the step of a subproblem comes
  // from its count-filter bound (SortedPoints::sub_bound, written in set_points), the conversion is exact to 32 bits
  int fx_patch;
  float fx_bound_limit;   // subproblems whose bound exceeds this go to the fp64-plane kernels (quantisation noise)
  float fx_tap[8];        // fx_patch: per-tap maxima of the fitted kernel (TapMax, the first w of them)
  // fx_patch plans whose spread runs over STACKS of tiles (spread_stack3_kernel; set per set_points) and the cutting
  // parameters when a debug call overrides the defaults (0: stack_params' rule)
  int stack;
  int stack_len, stack_cap;
  int fp64_stack;   // r06: double-precision 3-D plan on 16 x 16 x 4 (w = 7, 8) / 16 x 16 x 8 (w <= 6) tiles (fp64 planes): spread_wave3_stack_kernel may walk stacks
};
// OFF / ON pair of nufft_hip_options.tuning: -1 = by the plan's own rule, 0 = never, 1 = always
inline int tune_mode(const Geom& g, int off_bit, int on_bit) { return (g.tuning & on_bit) ? 1 : ((g.tuning & off_bit) ? 0 : -1); }

// Per-point record in tile-sorted order. float: 16 bytes, one dwordx4 access;
// the original point index shares the slot of the unused third coordinate in
// 1D/2D (3D keeps it in a side array). double: 32 bytes.
template <typename T> struct Rec;
template <> struct alignas(16) Rec<float> {
  uint32_t loc;              // stencil start relative to the tile: l0 | l1<<10 | l2<<20
  float z0, z1;              // Horner arguments in [-1, 1]
  union { float z2; int32_t idx; };
};
template <> struct alignas(16) Rec<double> {
  uint32_t loc;
  int32_t idx;
  double z0, z1, z2;
};

// 2-D float record of a plan whose strengths are known at sort time (one type-1 transform,
// nufft_hip_execute_with_points): same 16 bytes, the strength in place of z1 / idx and the
// positions packed as 5-bit tile-local start | 27-bit fixed-point Horner argument.
struct alignas(16) FusedRec {
  uint32_t px, py;           // l << 27 | round((z + 1) 2^26)
  float re, im;
};
// 3-D float counterpart (fixed-point plans on the ranked-scatter sort path, nufft_dense3.hip): the 16-byte
// record (index kept: the fp64-plane launches for crowded tiles still gather through it) followed by the
// strength, padded to 32 bytes -- ONE scattered 32-byte store per point in the sort (scattered stores are bound
// by transactions, not bytes: EXPERIMENTS.md section 5), and the spread kernel gathers nothing (the gather of 8-byte
// strengths through the sort permutation cost a 64-byte sector per point, twice: profiles/r03_pmc_cfg4.txt).
struct alignas(32) FusedRec3 {
  Rec<float> r;
  float re, im;
  uint32_t pad[2];
};
constexpr float kFusedScale = 67108864.0f;            // 2^26
constexpr float kFusedInv = 1.4901161193847656e-08f;  // 2^-26

template <typename T>
struct SortedPoints {
  const Rec<T>* rec;          // [M] tile-sorted
  const int32_t* tile_start;  // [ntiles + 1]
  const int32_t* sub_start;   // [ntiles + 1] exclusive scan of ceil(count / max_sub)
  // fixed-point 3-D float plans (nufft_dense3.hip), else null:
  float* cstats;              // [slots][2]: largest and summed max(|re c|, |im c|) of the strengths a launch spreads
                              // (written by launch_spread itself, before the spread kernel), then the partial pairs of
  int cstats_blocks;          // this many workgroups per slot (fixed by the plan from its largest slot count,
  int cstats_slots;           // cstats_slots: the partial pairs start behind that many result pairs)
  const float* sub_bound;     // Geom::fx_patch: [subproblem grid + 1] count-filter bound of every subproblem, negative =
                              // left to the fp64-plane kernels
  int* fb_ticket;             // (one int: the work counter of the persistent fp64-plane launch, zeroed before it)
  const int* fb_list;         // fixed-point plans: fb_list[0] = how many subproblems set_points left to the fp64-plane
                              // kernels, fb_list[1..] = their launch slots (bound3_kernel / crowded_list_kernel)
  // Geom::stack: the stacks stack_plan_kernel cut ({column, z0 | nz << 16, piece's points or -1}), how many, and
  // their count-filter bounds (negative: on the fallback list)
  const int4* segs;
  const int* seg_count;
  const float* seg_bound;
};
// Tap maxima of the fitted kernel, max over z of |P_t(z)| for every stencil cell t (with the fit's and the float
// evaluation's margin): what bound3_kernel filters the start-cell counts with
struct TapMax { float k[16]; };
template <typename T>
struct SortedOut {
  Rec<T>* rec;
};

struct PointsIn {
  const void* pts[3];   // x, y, z (unused dimensions alias x)
  int64_t stride;
  int64_t M;            // all points: nitems * M_item
  int64_t M_item;       // points per set (set k = points [k M_item, (k + 1) M_item))
  int blocks_per_item;  // sort workgroups per set (a workgroup never straddles two sets)
  int range_mode;
  int check_range;
  int aos;              // rank when pts[] are the columns of ONE [M, rank] array with x last (stride == rank), else 0
  const void* strengths;   // [M] interleaved complex: fused sort only (records carry them), else null
};

struct SortWork {
  int32_t* hist;        // [nblk][ntiles] (LDS-histogram path)
  int32_t* tile_of;     // [M] (global-counter path)
  int32_t* rank_of;     // [M]
  int32_t* tile_count;  // [ntiles]
  int32_t* tile_start;  // [ntiles + 1]
  int32_t* sub_start;   // [ntiles + 1]
  int32_t* bad_count;
  void* tmp;            // two-level sort: [M] 16-byte level-1 records
};
struct Sort2Layout { int64_t table, pieces, words; };
Sort2Layout sort2_layout(const Geom& g, int64_t M);

constexpr int kMaxLdsTiles = 16384;     // 64 KiB of 32-bit LDS counters
constexpr int kMaxLds16Tiles = 73728;   // 144 KiB of packed 16-bit LDS counters
constexpr int kMaxRanges16 = 4;         // tile ranges the 16-bit histogram pass may split into

enum Stage {
  STAGE_SORT_COUNT = 0, STAGE_SORT_SCAN, STAGE_SORT_SCATTER, STAGE_ZERO, STAGE_SPREAD,
  STAGE_FFT, STAGE_DECONVOLVE, STAGE_INTERP, STAGE_SORT_CELL, STAGE_COUNT
};
// Optional per-stage HIP-event timing (bench / profiling); no-ops when disabled.
struct StageHook {
  void* ctx = nullptr;
  void (*begin_fn)(void*, int) = nullptr;
  void (*end_fn)(void*, int) = nullptr;
  void begin(int s) const { if (begin_fn) begin_fn(ctx, s); }
  void end(int s) const { if (end_fn) end_fn(ctx, s); }
};

// Forces the (otherwise lazy) load of this library's device code and waits for it.
hipError_t preload_device_code();
// Launchers (nufft_kernels.hip). All enqueue on `stream` and return hipGetLastError().
// Workgroups of the LDS-histogram sorts: returns blocks PER ITEM (total = g.nitems * that);
// M is the total point count.
int sort_blocks(const Geom& g, int64_t M, int64_t* per_block);
int sort_blocks16(const Geom& g, int64_t M, int64_t* per_block);
int sort_mode(const Geom& g, int64_t M);
bool sort_uses_lds(const Geom& g);
template <typename T>
hipError_t launch_sort(const Geom& g, const PointsIn& in, const SortWork& w, const SortedOut<T>& out,
                       hipStream_t stream, const StageHook& hook);
// The fused sort (records carry the strengths) exists for this geometry / point count?
bool fused_sort_supported(const Geom& g, int method, int precision, int64_t M);
// Second sort level: reorders the records of every subproblem by stencil start cell,
// `in` -> `out` (distinct buffers). cellsort_wanted: the plan's spread kernel exploits the
// order (2-D, w = 8, float) and the point density makes it pay; the host applies it lazily
// (nufft_plan.cpp, maybe_cellsort).
bool cellsort_wanted(const Geom& g, int method, int precision, int64_t M);
bool cellsort_wanted_interp(const Geom& g, int method, int precision, int64_t M);
template <typename T>
hipError_t launch_cellsort(const Geom& g, int64_t M, const int32_t* tile_start, const int32_t* sub_start,
                           const Rec<T>* in, Rec<T>* out, hipStream_t stream);
template <typename T>
hipError_t launch_spread(const Geom& g, int method, const SortedPoints<T>& sp, int64_t M,
                         const T* horner, const T* c, T* fw, int batch, int64_t c_stride,
                         int64_t fw_stride, T scale, size_t lds_bytes, hipStream_t stream);
template <typename T>
hipError_t launch_interp(const Geom& g, int method, const SortedPoints<T>& sp, int64_t M,
                         const T* horner, T* c, const T* fw, int batch, int64_t c_stride,
                         int64_t fw_stride, T scale, hipStream_t stream);
// r06: interpolation straight from the caller's (unsorted) points, for small type-2 calls through the one-call entry
bool direct_interp_supported(const Geom& g);
template <typename T>
hipError_t launch_interp_direct(const Geom& g, const PointsIn& in, const T* horner, T* c, const T* fw, int batch,
                                int64_t c_stride, int64_t fw_stride, T scale, hipStream_t stream);
// dir 1: f = fw / phihat (type-1 step 3); dir 2: fw = f / phihat, zero elsewhere (type-2 step 1).
template <typename T>
hipError_t launch_deconvolve(const Geom& g, int dir, T* f, T* fw, const T* const rfser[3],
                             int batch, hipStream_t stream);
// Generic strided copy used by the op-level host code for batch-dimension permutes.
hipError_t launch_permute(const void* src, void* dst, int elem_bytes, int ndim,
                          const int64_t* out_shape, const int64_t* src_strides,
                          hipStream_t stream);
// Pruned per-dimension FFT passes with the deconvolution fused in (nufft_fft.hip): power-of-two
// fine grids up to 2048 per dimension. fine / f: [batch] fine grids / mode arrays; tmp0, tmp1:
// intermediates of pruned_fft_tmp_elems(g) complex elements per transform (tmp1: rank 3 only);
// tw[d]: exp(iflag 2 pi i m / nf_d), m < nf_d; rf[d]: reciprocal kernel Fourier series.
bool pruned_fft_supported(const Geom& g, int precision);
int64_t pruned_fft_tmp_elems(const Geom& g);
template <typename T>
hipError_t launch_pruned_fft(const Geom& g, int type, int iflag, T* fine, T* f, T* tmp0, T* tmp1,
                             const T* const rf[3], const T* const tw[3], int batch, hipStream_t stream,
                             bool zero_fine = false);   // type 1: leave `fine` zeroed (its first pass reads it all)
size_t spread_lds_bytes(const Geom& g, int method, int precision);
size_t interp_lds_bytes(const Geom& g, int method, int precision);
int wave_lstride(int rank);
// widths 9..16 (nufft_wide.hip): the tile the plan must sort by, its LDS row stride and the kernel's LDS bytes
bool wide_spread_supported(int rank, int w);
void wide_spread_tile(int rank, int w, int tile[3]);
int wide_spread_lstride(int rank, int w);
size_t wide_spread_lds_bytes(int rank, int w, int precision);
// 1-D interpolation (nufft_line.hip)
bool line_kernels_supported(const Geom& g);
size_t line_interp_lds_bytes(const Geom& g, int precision);
template <typename T>
hipError_t launch_interp_line(const Geom& g, const SortedPoints<T>& sp, int64_t M, const T* horner, T* c,
                              const T* fw, int batch, int64_t c_stride, int64_t fw_stride, T scale,
                              hipStream_t stream);
size_t wide_interp_lds_bytes(int rank, int w, int precision);
template <typename T>
hipError_t launch_interp_wide(const Geom& g, const SortedPoints<T>& sp, int64_t M, const T* horner, T* c,
                              const T* fw, int batch, int64_t c_stride, int64_t fw_stride, T scale,
                              hipStream_t stream);
template <typename T>
hipError_t launch_spread_wide(const Geom& g, const SortedPoints<T>& sp, int64_t M, const T* horner, const T* c,
                              T* fw, int batch, int64_t c_stride, int64_t fw_stride, T scale,
                              hipStream_t stream);
// 3-D float fixed-point spreading for w <= 6 (nufft_dense3.hip): lanes cover the stencil densely
bool dense3_supported(const Geom& g, int precision);
size_t dense3_lds_bytes(int w);
hipError_t launch_spread_dense3(const Geom& g, const SortedPoints<float>& sp, unsigned nsub_bound, int64_t M, const float* horner,
                                const float* c, float* fw, int batch, int64_t c_stride, int64_t fw_stride, float scale,
                                hipStream_t stream);
// w = 7, 8 (same file): exact-conversion fixed point on 8 x 8 (x, y) lane patches, step from the count-filter bound
bool patch3_supported(const Geom& g, int precision);
size_t patch3_lds_bytes(int w);
hipError_t launch_spread_patch3(const Geom& g, const SortedPoints<float>& sp, unsigned nsub_bound, const float* horner,
                                const float* c, float* fw, int batch, int64_t c_stride, int64_t fw_stride, float scale,
                                hipStream_t stream);
// set_points of such a plan: sub_bound[s] for every subproblem s of the launch grid, fb_list = those left to the
// fp64-plane kernels (count zeroed here); rec_stride: bytes between records
hipError_t launch_bound3(const Geom& g, const Rec<float>* rec, int rec_stride, const int32_t* tile_start, const int32_t* sub_start,
                         unsigned nsub_bound, const TapMax& taps, float* sub_bound, int* fb_list, hipStream_t stream);
// stacks of tiles (r05, same file): which plans take them, the launch grid bound, cutting, bounds, spreading
bool stack3_wanted(const Geom& g, int64_t M);
void stack_params(const Geom& g, int* cap, int* len);
unsigned stack_grid_bound(const Geom& g, int64_t M);
hipError_t launch_stack_plan(const Geom& g, const int32_t* tile_start, int64_t M, int4* segs, int* seg_count, hipStream_t stream);
hipError_t launch_bound3_stack(const Geom& g, const Rec<float>* rec, int rec_stride, const int32_t* tile_start,
                               const int32_t* sub_start, int64_t M, const TapMax& taps, const int4* segs, const int* seg_count,
                               float* seg_bound, int* fb_list, hipStream_t stream);
hipError_t launch_spread_dense3_stack(const Geom& g, const SortedPoints<float>& sp, int64_t M, const float* horner,
                                      const float* c, float* fw, int batch, int64_t c_stride, int64_t fw_stride, float scale,
                                      hipStream_t stream);
hipError_t launch_spread_stack3(const Geom& g, const SortedPoints<float>& sp, int64_t M, const float* horner,
                                const float* c, float* fw, int batch, int64_t c_stride, int64_t fw_stride, float scale,
                                hipStream_t stream);
// the other fixed-point plans: fb_list from the tiles with more than fx_max_subs subproblems
hipError_t launch_crowded_list(const Geom& g, const int32_t* sub_start, int* fb_list, hipStream_t stream);
// the cell-grouped fp64-plane launch for the subproblems on fb_list of a w = 7, 8 fixed-point plan (nufft_dense3.hip)
hipError_t launch_spread_group3_fallback(const Geom& g, const SortedPoints<float>& sp, unsigned nsub_bound, const float* horner,
                                         const float* c, float* fw, int batch, int64_t c_stride, int64_t fw_stride, float scale,
                                         hipStream_t stream);
// strengths of one spread launch: cstats[slot] = {max, sum} of max(|re c|, |im c|) over the slot's M points;
// the buffer holds cstats_floats(M, slots) floats (the results, then per-workgroup partial pairs)
hipError_t launch_cstats(const float* c, int64_t M, int slots, int nblk, int max_slots, int64_t c_stride, float* cstats, hipStream_t stream);
int cstats_blocks(int64_t M, int max_slots);                 // workgroups per slot
size_t cstats_floats(int64_t M, int max_slots);
unsigned subproblem_grid_bound(const Geom& g, int64_t M);   // launch grid of the subproblem kernels (>= live subproblems)
hipError_t measure_shader_clock_mhz(hipStream_t stream, unsigned long long* scratch, double* mhz);
int wave3_pad(int w);   // spill elements behind the LDS planes of the 3-D wavefront kernel
bool sparse_wanted(const Geom& g, int64_t M);   // point set sparse enough for the LDS-free spreader
bool wave_method_supported(const Geom& g, int precision);

}  // namespace nufft_hip

#endif  // NUFFT_HIP_INTERNAL_H_
