/* include/nufft_hip.h -- C ABI of libnufft_hip.so, the MI355X (gfx950) NUFFT
 * plan library. Plain pointers and sizes only: no TensorFlow, torch or C++
 * types cross this boundary.
 *
 * It replaces, for the GPU device, the interface the reference's TensorFlow
 * op kernels call (paths relative to the reference checkout,
 * tensorflow_nufft/cc/kernels/):
 *
 *   Plan<GPUDevice,FloatType>::initialize   nufft_plan.h:223-231, nufft_plan.cu.cc:1808-2030
 *   Plan<GPUDevice,FloatType>::set_points   nufft_plan.h:237-241, nufft_plan.cu.cc:2054-2111
 *   Plan<GPUDevice,FloatType>::execute      nufft_plan.h:245,     nufft_plan.cu.cc:2113-2168
 *   Plan<GPUDevice,FloatType>::interp       nufft_plan.h:250,     nufft_plan.cu.cc:2170-2196
 *   Plan<GPUDevice,FloatType>::spread       nufft_plan.h:255,     nufft_plan.cu.cc:2198-2225
 *   NUFFTBaseOp::Compute / ::Execute        nufft_kernels.cc:54-542 (op-level entry below)
 *
 * Conventions (same as the reference plan):
 *  - grid dimensions are given x-fastest ("FINUFFT order" = the TF grid shape
 *    reversed, nufft_kernels.cc:347-352); mode arrays f are [ntransf][N3][N2][N1]
 *    contiguous with N1 fastest, CMCL mode order (index 0 = most negative mode);
 *  - strengths c are [ntransf][M] interleaved complex;
 *  - points are device arrays of the plan's real type, in radians/sample;
 *  - iflag -1 = 'forward' (exp(-i k x)), +1 = 'backward' (nufft_plan.h:126-129);
 *  - every pointer marked "device" must be valid on the plan's device; all work
 *    is enqueued on the plan's stream and no call synchronises the device
 *    unless stated.
 * Unlike the reference, the points buffer is NOT modified (the reference folds
 * and rescales it in place, nufft_plan.h:902-947).
 *
 * Thread safety: distinct plans may be used concurrently from distinct
 * threads; one plan must not be used from two threads at once (the reference
 * builds one Plan per Compute call, nufft_kernels.cc:474-475).
 */
#ifndef NUFFT_HIP_H_
#define NUFFT_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history (additive since 1: a host built against version n runs on any library >= n and can ask
 * nufft_hip_abi_version() for what it needs):
 *   1  plan_create / set_points / execute / spread / interp / destroy, plan_get_info, last_error, default_options,
 *      op_shape / op_compute
 *   2  plan_create_host, plan_create_ex with allocator callbacks, plan_set_allocator, plan_release_workspace,
 *      op_compute_ex, execute_with_points, plan_set_stream, timing, plan cache controls, debug_stop_after /
 *      copy_fine_grid / fseries / eval_kernel / sort_path, plans with several point sets
 *   3  options.tuning (validated bits), options_from_proto, op_desc_from_attrs, plan_describe
 *   4  (r06) nufft_hip_build_info. Entries that arrived under version 3's number in r04 / r05 and are guaranteed from
 *      4 on: debug_sub_bounds, debug_shader_clock_mhz, debug_stacks, debug_stack_params; tuning bits FXPATCH_OFF,
 *      QFOLD_OFF, STACK_OFF / STACK_ON, FBGROUP_OFF, and r06's MIXFFT_OFF, ISPLIT_OFF / ISPLIT_ON, DIRECT_OFF / DIRECT_ON */
#define NUFFT_HIP_ABI_VERSION 4

/* Status codes. They map onto the tensorflow::errors the reference returns. */
enum {
  NUFFT_HIP_OK = 0,
  NUFFT_HIP_INVALID_ARGUMENT = 3,    /* errors::InvalidArgument  */
  NUFFT_HIP_RESOURCE_EXHAUSTED = 8,  /* errors::ResourceExhausted */
  NUFFT_HIP_UNIMPLEMENTED = 12,      /* errors::Unimplemented    */
  NUFFT_HIP_INTERNAL = 13            /* errors::Internal (HIP / rocFFT failure) */
};

enum { NUFFT_HIP_TYPE_1 = 1, NUFFT_HIP_TYPE_2 = 2 };
enum { NUFFT_HIP_FORWARD = -1, NUFFT_HIP_BACKWARD = 1 };
enum { NUFFT_HIP_F32 = 4, NUFFT_HIP_F64 = 8 };          /* bytes of the real type */
/* PointsRange, tensorflow_nufft/proto/nufft_options.proto and nufft_plan.h:676-734 */
enum { NUFFT_HIP_RANGE_STRICT = 0, NUFFT_HIP_RANGE_EXTENDED = 1, NUFFT_HIP_RANGE_INFINITE = 2 };
/* spread_method: how type-1 spreading / type-2 interpolation run */
enum {
  NUFFT_HIP_METHOD_AUTO = 0,
  NUFFT_HIP_METHOD_TILE_GENERIC = 1, /* LDS tile, one thread per point (any w, rank, precision) */
  NUFFT_HIP_METHOD_TILE_WAVE = 2,    /* LDS tile, one point per wavefront pass (specialised widths) */
  NUFFT_HIP_METHOD_POINT_GLOBAL = 3  /* no LDS tile: every point adds its stencil to the fine grid with
                                        global atomics (the reference's nupts-driven method,
                                        nufft_plan.cu.cc:2325-2436); AUTO picks it for sparse point sets */
};
enum { NUFFT_HIP_OP_NUFFT = 0, NUFFT_HIP_OP_INTERP = 1, NUFFT_HIP_OP_SPREAD = 2 };

/* Options. The first four fields are the reference's user-visible Options
 * (python/ops/nufft_options.py:222-273, proto/nufft_options.proto:27-32); the
 * rest correspond to InternalOptions (nufft_options.h:92-162). Zero means
 * "choose automatically" everywhere. Initialise with nufft_hip_default_options. */
typedef struct nufft_hip_options {
  int32_t max_batch_size;      /* Options.max_batch_size; 0 = auto */
  int32_t points_range;        /* NUFFT_HIP_RANGE_*; default EXTENDED */
  int32_t check_points_range;  /* debugging.check_points_range (costs one host sync) */
  int32_t fftw_planning_rigor; /* accepted, ignored (rocFFT) */
  int32_t spread_only;         /* Interp / Spread ops: no upsampling, no FFT */
  int32_t kernel_width;        /* 0 = from tol */
  double upsampling_factor;    /* 0 = 2.0 */
  int32_t spread_method;       /* NUFFT_HIP_METHOD_* */
  int32_t max_subproblem_size; /* points per workgroup pass; 0 = auto */
  int32_t tile_dims[3];        /* fine-grid cells per tile, x fastest; 0 = auto */
  int32_t lds_accumulate;      /* LDS tile accumulation: 0 auto, 1 double, 2 packed 32+32-bit fixed
                                  point (3-D float, kernel width <= 8 only; tiles holding more than 16
                                  subproblems, and at widths 7, 8 subproblems whose count-filter bound makes
                                  the step too coarse, are accumulated in double all the same) */
  int32_t num_point_sets;      /* K > 1: the plan sorts and transforms K independent point sets at once
                                  (batched transforms with per-item points, nufft_kernels.cc:491-540 run as
                                  one pass): set_points takes K * num_points points, set k at
                                  [k num_points, (k+1) num_points); c is [K][ntransf][M], f [K][ntransf][grid];
                                  every transform of every set is in flight together (one fine grid each) */
  int32_t tuning;              /* NUFFT_HIP_TUNE_* bits: forces one of the kernel families that the plan otherwise
                                  chooses between by point density / geometry (second-opinion tests, A/B
                                  measurements). Results do not depend on it beyond rounding. 0 = automatic */
  int32_t op_group;            /* op-level entry: point sets handled per plan call; 0 = auto (16), 1 = one at a time */
  int32_t op_lanes;            /* op-level entry: plans pipelined on private streams, 1..4; 0 = auto (2) */
  int32_t reserved[3];
} nufft_hip_options;

/* nufft_hip_options.tuning (every OFF / ON pair: neither bit = by the plan's density / geometry rule) */
enum {
  NUFFT_HIP_TUNE_NO_FUSED = 1 << 0,        /* execute_with_points never sorts the strengths into the records */
  NUFFT_HIP_TUNE_GROUP_OFF = 1 << 1,       /* 2-D cell-grouped spreader (spread_2d_w8_group_kernel): never / always */
  NUFFT_HIP_TUNE_GROUP_ON = 1 << 2,
  NUFFT_HIP_TUNE_SPARSE_OFF = 1 << 3,      /* LDS-free spreader under spread_method AUTO: never / always */
  NUFFT_HIP_TUNE_SPARSE_ON = 1 << 4,
  NUFFT_HIP_TUNE_CELLSORT_OFF = 1 << 5,    /* 2-D: records reordered by start cell on plan reuse: never / always */
  NUFFT_HIP_TUNE_CELLSORT_ON = 1 << 6,
  NUFFT_HIP_TUNE_CELLSORT3D_OFF = 1 << 7,  /* 3-D type 2: records ordered by start cell in set_points: never / always */
  NUFFT_HIP_TUNE_CELLSORT3D_ON = 1 << 8,
  NUFFT_HIP_TUNE_ROCFFT = 1 << 9,          /* rocFFT + deconvolve kernel instead of the pruned FFT passes */
  NUFFT_HIP_TUNE_NO_WIDE = 1 << 10,        /* w = 9..16 on the thread-per-point tile kernels */
  NUFFT_HIP_TUNE_NO_LINE = 1 << 11,        /* 1-D type 2 on the gather-from-global kernel */
  NUFFT_HIP_TUNE_JOINT_OFF = 1 << 12,      /* 3-D float w = 8: both fp64 planes in one launch: never / always */
  NUFFT_HIP_TUNE_JOINT_ON = 1 << 13,
  NUFFT_HIP_TUNE_STAGED_OFF = 1 << 14,     /* staged scatter (<= 1024 tiles per point set): never / always */
  NUFFT_HIP_TUNE_STAGED_ON = 1 << 15,
  NUFFT_HIP_TUNE_SORT2_OFF = 1 << 16,      /* 3-D float: two-level sort (64^3-cell super-tiles first): never / wherever it exists.
                                              Its level-1 records keep the Horner argument to 2^-25 (the one-level sorts: 2^-27),
                                              so switching it moves a point by up to 1.5e-8 of a fine cell -- below what a float
                                              argument resolves near |z| = 1 (6e-8), but not bit-identical */
  NUFFT_HIP_TUNE_SORT2_ON = 1 << 17,
  NUFFT_HIP_TUNE_FXPATCH_OFF = 1 << 18,    /* 3-D float w = 7, 8: the r03 kernels (fp64 planes at w = 8; w = 7 fixed point on
                                              depth-4 tiles, 512-point subproblems) instead of spread_patch3_kernel */
  NUFFT_HIP_TUNE_QFOLD_OFF = 1 << 19,      /* LDS-histogram sorts: the general coordinate fold (fmod / division / 64-bit modulo
                                              paths compiled in) although the plan's tiles are powers of two and the points'
                                              range is STRICT or EXTENDED -- same results, for A/B runs */
  NUFFT_HIP_TUNE_STACK_OFF = 1 << 20,      /* 3-D spreading: one workgroup per STACK of tiles consecutive in z, the z halo carried in
                                              LDS instead of written out per tile (r05, float fixed point: by default below 0.22 /
                                              0.25 / 1.0 points per fine cell at w = 7, 8 / 5, 6 / <= 4; r06, double precision w <= 8:
                                              below 2.0, w = 9..16 either precision: below 0.5): never / always */
  NUFFT_HIP_TUNE_STACK_ON = 1 << 21,
  NUFFT_HIP_TUNE_FBGROUP_OFF = 1 << 22,    /* 3-D float fixed-point plans: the subproblems left to the fp64 planes (bound above the limit,
                                              crowded tiles) on the r04 kernel (one launch per component, an atomic per point and
                                              plane) instead of the cell-grouped one (r05) */
  NUFFT_HIP_TUNE_MIXFFT_OFF = 1 << 23,     /* fine-grid dimensions that are not powers of two (or exceed 2048): rocFFT + deconvolve kernel
                                              instead of the mixed-radix pruned passes (r06); power-of-two grids keep their passes */
  NUFFT_HIP_TUNE_ISPLIT_OFF = 1 << 24,     /* 3-D interpolation: eight lanes per point (each sums one z plane of the stencil) instead of
                                              a thread per point (r06; by default at w = 7, 8 while a tile holds < 64 points on average): never / always */
  NUFFT_HIP_TUNE_ISPLIT_ON = 1 << 25,
  NUFFT_HIP_TUNE_DIRECT_OFF = 1 << 26,     /* type 2 through the one-call entry (execute_with_points, what an op invocation runs): interpolate
                                              straight from the caller's unsorted points, no sort (r06; by default for small calls: 2-D up to
                                              1e5 points, 3-D while a point has > 256 fine cells to itself): never / always */
  NUFFT_HIP_TUNE_DIRECT_ON = 1 << 27,
  NUFFT_HIP_TUNE_ALL = (1 << 28) - 1       /* every defined bit: plan creation refuses others, and both bits of a pair */
};

typedef struct nufft_hip_plan_s* nufft_hip_plan;

/* Device-memory allocator supplied by the host framework. The reference plan
 * takes its fine grid and Fourier-series buffers from
 * OpKernelContext::allocate_temp (nufft_plan.cu.cc:1981-1986) and the sort
 * arrays from the TF device allocator (:2032-2052, :2927-2940); a TensorFlow
 * binding routes `alloc` to allocate_temp and lets the tensors die with the
 * Compute call (`free` may be a no-op). Both callbacks must be stream-safe in
 * the host framework's sense: memory handed back by `free` is not reused before
 * the work already enqueued on the plan's stream has finished.
 * A NULL allocator (or NULL members) means hipMalloc / hipFree. */
typedef struct nufft_hip_allocator {
  void* (*alloc)(size_t bytes, void* user);   /* 256-byte aligned device memory, NULL on failure */
  void (*free)(void* ptr, void* user);
  void* user;
} nufft_hip_allocator;

typedef struct nufft_hip_plan_info {
  int32_t type, rank, precision, iflag, ntransf, batch_size;
  int32_t kernel_width, ncoef, spread_method;
  double upsampling_factor, beta, tol;
  int64_t grid_dims[3], fine_dims[3];
  int32_t tile_dims[3], num_tiles[3];
  int32_t max_subproblem_size;
  int64_t num_points;
  int64_t workspace_bytes;
} nufft_hip_plan_info;

int nufft_hip_abi_version(void);
/* What this binary is, as one line of key=value pairs separated by ';':
 * "abi=4;source=<16 hex digits: SHA-256 over the library's sources>;arch=gfx950;experiment=none". A library built with
 * any of the experiment macros of csrc/nufft_experiment.h (kernel shapes changed, or pieces of a main loop left out
 * for timing: wrong results) reports them after "experiment=" -- a host or test can refuse it. Static storage. (ABI 4) */
const char* nufft_hip_build_info(void);
void nufft_hip_default_options(nufft_hip_options* opts);

/* = Plan::initialize. grid_dims has `rank` entries, x fastest. `stream` is a
 * hipStream_t (NULL = the null stream). On failure *plan is NULL and, if
 * errbuf is given, a message is written there. */
int nufft_hip_plan_create(nufft_hip_plan* plan, int type, int rank,
                          const int64_t* grid_dims, int iflag, int ntransf,
                          double tol, int precision,
                          const nufft_hip_options* opts, void* stream,
                          char* errbuf, size_t errbuf_len);

/* The same with the plan's WORKSPACE (fine grid, sorted records, sort tables,
 * FFT work buffer) taken from `allocator`. The small constant tables (kernel
 * polynomial, Fourier-series reciprocals) and rocFFT's own twiddles stay
 * internal. nufft_hip_plan_release_workspace hands every workspace buffer back
 * (the plan forgets its points; the next set_points allocates again), so a
 * cached plan can outlive the framework's per-call temporaries;
 * nufft_hip_plan_set_allocator swaps the allocator of a plan that currently
 * holds no workspace (e.g. the `user` pointer is a per-call context). */
int nufft_hip_plan_create_ex(nufft_hip_plan* plan, int type, int rank,
                             const int64_t* grid_dims, int iflag, int ntransf,
                             double tol, int precision,
                             const nufft_hip_options* opts, void* stream,
                             const nufft_hip_allocator* allocator,
                             char* errbuf, size_t errbuf_len);
int nufft_hip_plan_release_workspace(nufft_hip_plan plan);
int nufft_hip_plan_set_allocator(nufft_hip_plan plan, const nufft_hip_allocator* allocator);
/* Host-only plan: every parameter rule, the kernel polynomial and the Fourier
 * series, no device state (works without a GPU). Usable with get_info,
 * debug_fseries, debug_eval_kernel and destroy only. */
int nufft_hip_plan_create_host(nufft_hip_plan* plan, int type, int rank,
                               const int64_t* grid_dims, int iflag, int ntransf,
                               double tol, int precision,
                               const nufft_hip_options* opts,
                               char* errbuf, size_t errbuf_len);

/* = Plan::set_points. x, y, z: device pointers (y, z ignored below rank 2, 3).
 * `stride` is the element stride between consecutive points (1 for separate
 * arrays; `rank` when x, y, z point into one [M, rank] array). */
int nufft_hip_set_points(nufft_hip_plan plan, int64_t num_points,
                         const void* x, const void* y, const void* z,
                         int64_t stride);

/* = Plan::execute. Type 1: reads c, writes f. Type 2: reads f, writes c. */
int nufft_hip_execute(nufft_hip_plan plan, void* c, void* f);
/* set_points followed by execute, as ONE call: what a NUFFT op invocation does
 * (nufft_kernels.cc:491-540). Knowing the strengths at sort time lets a type-1
 * plan with one transform carry them inside the sorted records (no gather in
 * the spread kernel). The points are consumed: a later nufft_hip_execute on the
 * same plan needs a new nufft_hip_set_points. */
int nufft_hip_execute_with_points(nufft_hip_plan plan, int64_t num_points,
                                  const void* x, const void* y, const void* z,
                                  int64_t stride, void* c, void* f);
/* = Plan::spread / Plan::interp (plan created with opts.spread_only = 1):
 * f is the [ntransf][grid] array itself (no upsampling). */
int nufft_hip_spread(nufft_hip_plan plan, const void* c, void* f);
int nufft_hip_interp(nufft_hip_plan plan, void* c, const void* f);

int nufft_hip_plan_get_info(nufft_hip_plan plan, nufft_hip_plan_info* info);
/* Host-only: runs every parameter rule of plan creation (kernel width, fine
 * grid, tiles, polynomial degree, batch size, method) without touching a
 * device, and reports the result. Same arguments / errors as plan_create. */
int nufft_hip_plan_describe(int type, int rank, const int64_t* grid_dims, int iflag,
                            int ntransf, double tol, int precision,
                            const nufft_hip_options* opts, nufft_hip_plan_info* info,
                            char* errbuf, size_t errbuf_len);
int nufft_hip_plan_set_stream(nufft_hip_plan plan, void* stream);
/* Per-stage timing with HIP events recorded on the plan's stream around each
 * stage (no synchronisation while enabled). get_timing synchronises the
 * stream, returns accumulated milliseconds and call counts per stage since the
 * last get, and resets them. Stage order: sort-count, sort-scan, sort-scatter,
 * zero, spread, fft, deconvolve, interp, sort-cell (the per-subproblem ordering
 * by stencil start cell, when the plan uses it). enable: 0 off, 1 every stage, 2 only
 * the dominant kernel (spread / interp; two events per execute). */
#define NUFFT_HIP_NUM_STAGES 9
int nufft_hip_plan_set_timing(nufft_hip_plan plan, int enable);
int nufft_hip_plan_get_timing(nufft_hip_plan plan, double* ms, int32_t* calls, int n);
const char* nufft_hip_last_error(nufft_hip_plan plan);
int nufft_hip_plan_destroy(nufft_hip_plan plan);

/* Debug/test access to intermediate stages (device pointers, valid until the
 * next call on the plan): the fine grid of the last batch, and the kernel
 * Fourier-series reciprocals per dimension (host copy). */
int nufft_hip_debug_fine_grid(nufft_hip_plan plan, void** fine, int64_t* count);
/* Copies `count` complex elements of the fine grid to the device buffer dst (on the plan's stream). */
int nufft_hip_debug_copy_fine_grid(nufft_hip_plan plan, void* dst, int64_t count);
int nufft_hip_debug_fseries(nufft_hip_plan plan, int dim, double* out, int64_t n);
/* Evaluates the plan's piecewise-polynomial kernel on the host for n offsets
 * x1 in [-w/2, -w/2+1] (out: n*w values, normalised so that phi(0) = 1). */
int nufft_hip_debug_eval_kernel(nufft_hip_plan plan, int n, const double* x1, double* out);
/* Makes execute return right after the given stage (index as in
 * nufft_hip_plan_get_timing: 4 = spread, 5 = fft, 6 = deconvolve/amplify), so
 * that tests can compare the fine grid stage by stage; -1 = run everything. */
int nufft_hip_debug_stop_after(nufft_hip_plan plan, int stage);
/* Which sort the plan ran at its last set_points: 0 = LDS histogram, 1 = 16-bit LDS histogram + ranked scatter,
 * 2 = global counters, 3 = two levels (super-tiles, then tiles); -1 = no points set. */
int nufft_hip_debug_sort_path(nufft_hip_plan plan);
/* 3-D float plans at kernel widths 7, 8 (packed fixed point, r04): the count-filter bound of every subproblem as the
 * last set_points wrote it, host copy (synchronises): out[s] > 0 = the bound B of subproblem s (every cell sum of its
 * spread is at most B largest strengths), < 0 = left to the double-precision LDS planes, 0 = unused launch slot.
 * Returns the number of launch slots (<= n copied), 0 for other plans, negative on error. */
int64_t nufft_hip_debug_sub_bounds(nufft_hip_plan plan, float* out, int64_t n);
/* 3-D plans spreading over STACKS of tiles (r05: float fixed-point plans; r06: double-precision plans and w = 9..16;
 * options.tuning STACK_OFF for the per-subproblem form). Float w = 7, 8: nufft_hip_debug_sub_bounds then reports one
 * bound per stack (no unused slots); float w <= 6, double precision and w = 9..16 cut stacks without per-stack bounds
 * (sub_bounds returns 0 entries there). This entry returns the stacks themselves, four int32 each: {tile column (t0 + ntile0 * t1, + columns per point set * set), z0 | nz << 16 = first
 * tile in z and how many, p0, p1 = the point range of a piece of a tile with more than max_subproblem_size points,
 * -1 otherwise}. Returns the number of stacks (<= n copied), 0 for other plans, negative on error. Synchronises. */
int64_t nufft_hip_debug_stacks(nufft_hip_plan plan, int32_t* out, int64_t n);
/* Overrides the cutting rule of the next set_points: at most `len` tiles and `cap` points per stack (0: the
 * plan's own rule). For A/B runs; the plan needs set_points again. */
int nufft_hip_debug_stack_params(nufft_hip_plan plan, int len, int cap);
/* Shader clock of the current device in MHz, measured while every CU runs LDS atomics for about a millisecond
 * (ratio of the shader-cycle counter to the constant-rate wall clock inside the kernel): what the LDS roofline of
 * bench.py is priced at -- the nominal 2400 MHz is not what these kernels run at. Allocates, launches on `stream`,
 * synchronises. */
int nufft_hip_debug_shader_clock_mhz(void* stream, double* mhz);

/* ---- Op-level entry: the host logic of NUFFTBaseOp::Compute/Execute --------
 * (nufft_kernels.cc:54-542): validation with the reference's error messages,
 * batch-shape broadcasting (dimensions where the points batch is 1 become
 * num_transforms; the others become sequential set_points+execute calls),
 * and the loop over calls. Shapes are in TensorFlow order (grid slowest first,
 * points [..., M, rank] with the last axis ordered like the grid).
 *
 * Call nufft_hip_op_shape first to validate and obtain the output shape,
 * allocate `target` (device), then nufft_hip_op_compute.                   */
typedef struct nufft_hip_op_desc {
  int32_t op_type;         /* NUFFT_HIP_OP_* */
  int32_t transform_type;  /* NUFFT_HIP_TYPE_* */
  int32_t fft_direction;   /* NUFFT_HIP_FORWARD / BACKWARD */
  int32_t precision;       /* NUFFT_HIP_F32 / F64 */
  double tol;              /* the op's `tol: float` attr (cc/ops/nufft_ops.cc:214) */
  nufft_hip_options options;
  int32_t source_ndim, points_ndim, grid_shape_len;
  int64_t source_shape[12];
  int64_t points_shape[12];
  int64_t grid_shape[3];   /* the `grid_shape` input (type 1; TF order) */
} nufft_hip_op_desc;

/* Decodes the op's `options` attr: the bytes of a serialized
 * tensorflow.nufft.Options message (proto/nufft_options.proto:19-32), which the
 * reference kernel parses with Options::ParseFromString (nufft_kernels.cc:582-585).
 * proto3 rules: absent fields take their defaults (points_range absent = STRICT,
 * the enum's zero; the Python wrapper always sends its own default EXTENDED),
 * unknown fields are skipped, the last value of a repeated scalar wins. Fills
 * the four user-visible fields of *out and sets every other field to its
 * nufft_hip_default_options value. Returns NUFFT_HIP_INVALID_ARGUMENT for bytes
 * the protobuf runtime would refuse (the reference then fails with
 * InvalidArgument "Unable to parse options string."); *out is left untouched.
 * Host only: no device is needed. */
int nufft_hip_options_from_proto(const void* bytes, size_t n, nufft_hip_options* out);
/* Fills *desc from the op's attrs exactly as the reference op constructors do
 * (NUFFT: nufft_kernels.cc:559-585; Interp :590-604; Spread :607-621):
 * transform_type 'type_1' / 'type_2', fft_direction 'forward' / 'backward'
 * (both ignored for Interp and Spread, may be NULL), the `tol: float` attr, the
 * serialized options (NUFFT only). Shapes are zeroed; the caller fills them per
 * Compute call. Host only. */
int nufft_hip_op_desc_from_attrs(nufft_hip_op_desc* desc, int op_type,
                                 const char* transform_type, const char* fft_direction,
                                 double tol, int precision,
                                 const void* options, size_t options_len,
                                 char* errbuf, size_t errbuf_len);

int nufft_hip_op_shape(const nufft_hip_op_desc* desc, int32_t* target_ndim,
                       int64_t* target_shape /* >= 12 entries */,
                       char* errbuf, size_t errbuf_len);
int nufft_hip_op_compute(const nufft_hip_op_desc* desc, const void* source,
                         const void* points, void* target, void* stream,
                         char* errbuf, size_t errbuf_len);
/* The same with the host framework's allocator: the plan workspace and the
 * batch-permute temporaries come from it and are handed back before the call
 * returns (cached plans keep only their constant tables and FFT plans). */
int nufft_hip_op_compute_ex(const nufft_hip_op_desc* desc, const void* source,
                            const void* points, void* target, void* stream,
                            const nufft_hip_allocator* allocator,
                            char* errbuf, size_t errbuf_len);
/* Plans built by nufft_hip_op_compute are cached per configuration (least
 * recently used first out, at most 16 plans and `max_bytes` of device memory:
 * default 8 GiB, nufft_hip_op_set_cache_limit changes it); clear_cache
 * releases them and their device memory. */
void nufft_hip_op_clear_cache(void);
void nufft_hip_op_set_cache_limit(int64_t max_bytes);
int64_t nufft_hip_op_cache_bytes(void);

#ifdef __cplusplus
}
#endif
#endif  /* NUFFT_HIP_H_ */
