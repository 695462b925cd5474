// Op-level host logic behind nufft_hip_op_shape / nufft_hip_op_compute: what
// the reference's TensorFlow kernel does around the plan
// (tensorflow_nufft/cc/kernels/nufft_kernels.cc, NUFFTBaseOp::Compute :54-379
// and ::Execute :381-542; shape function cc/ops/nufft_ops.cc:27-103), restated
// over plain shapes and device pointers so that the TF glue, the ctypes
// binding and any other host can share it.
//
// Differences by design: points are consumed in their [.., M, rank]
// layout through a strided read instead of being reversed and transposed into
// a temporary (reference :276-303); plans are cached per configuration instead
// of being rebuilt on every call (reference :474-478).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <list>
#include <mutex>
#include <numeric>
#include <map>
#include <string>
#include <vector>

#include "nufft_hip_internal.h"

namespace {

using nufft_hip::launch_permute;

std::string shape_str(const int64_t* s, int n) {   // TensorShape::DebugString format
  std::string r = "[";
  for (int i = 0; i < n; ++i) {
    if (i) r += ",";
    r += std::to_string((long long)s[i]);
  }
  return r + "]";
}

struct Analysis {
  int rank = 0;
  int64_t num_points = 0;
  std::vector<int64_t> grid;            // TF order
  std::vector<int64_t> source_batch;    // padded to common batch rank
  std::vector<int64_t> points_batch;
  std::vector<int64_t> out_batch;       // broadcast
  std::vector<int> outer, inner;        // batch dims with points dim != 1 / == 1
  int64_t num_transforms = 1, num_calls = 1;
  int source_elem_rank = 1;
  bool transpose = false;
  std::vector<int64_t> target_shape;
};

int fail(char* errbuf, size_t n, int code, const std::string& msg) {
  if (errbuf && n) snprintf(errbuf, n, "%s", msg.c_str());
  return code;
}

// Validation + shape algebra of NUFFTBaseOp::Compute (nufft_kernels.cc:58-274)
// and NUFFTBaseShapeFn (nufft_ops.cc:27-103). Error strings are the reference's.
int analyze(const nufft_hip_op_desc* d, Analysis* a, std::string* err) {
  if (d->points_ndim < 2 || d->points_ndim > 12 || d->source_ndim < 1 || d->source_ndim > 12) {
    *err = "Input `points` must have rank of at least 2, but got shape: " +
           shape_str(d->points_shape, std::max(0, d->points_ndim));
    return NUFFT_HIP_INVALID_ARGUMENT;
  }
  const int64_t rank = d->points_shape[d->points_ndim - 1];
  if (rank < 1 || rank > 3) {
    *err = "Dimension must be 1, 2 or 3, but is " + std::to_string((long long)rank);
    return NUFFT_HIP_INVALID_ARGUMENT;
  }
  a->rank = (int)rank;
  a->num_points = d->points_shape[d->points_ndim - 2];
  const bool t1 = d->transform_type == NUFFT_HIP_TYPE_1;
  if (d->transform_type != NUFFT_HIP_TYPE_1 && d->transform_type != NUFFT_HIP_TYPE_2) {
    *err = "transform_type attr must be 'type_1' or 'type_2'";
    return NUFFT_HIP_INVALID_ARGUMENT;
  }
  if (t1) {
    if (d->grid_shape_len != rank) {
      *err = "grid_shape must have length " + std::to_string((long long)rank) + " for a " +
             std::to_string((long long)rank) + "D transform (as inferred from points), but got length: " +
             std::to_string(d->grid_shape_len);
      return NUFFT_HIP_INVALID_ARGUMENT;
    }
    for (int i = 0; i < rank; ++i) {
      if (d->grid_shape[i] < 0) {
        *err = "Dimension " + std::to_string((long long)d->grid_shape[i]) + " must be >= 0";
        return NUFFT_HIP_INVALID_ARGUMENT;
      }
      a->grid.push_back(d->grid_shape[i]);
    }
    if (d->source_shape[d->source_ndim - 1] != a->num_points) {
      *err = "source and points must have equal samples dimensions for type-1 transforms, but got "
             "source.shape[-1] = " + std::to_string((long long)d->source_shape[d->source_ndim - 1]) +
             " and points.shape[-2] = " + std::to_string((long long)a->num_points);
      return NUFFT_HIP_INVALID_ARGUMENT;
    }
    a->source_elem_rank = 1;
  } else {
    if (d->source_ndim < rank) {
      *err = "Input `source` must have rank of at least " + std::to_string((long long)rank) +
             " but received shape: " + shape_str(d->source_shape, d->source_ndim);
      return NUFFT_HIP_INVALID_ARGUMENT;
    }
    for (int i = d->source_ndim - (int)rank; i < d->source_ndim; ++i) a->grid.push_back(d->source_shape[i]);
    a->source_elem_rank = (int)rank;
  }
  std::vector<int64_t> sb(d->source_shape, d->source_shape + d->source_ndim - a->source_elem_rank);
  std::vector<int64_t> pb(d->points_shape, d->points_shape + d->points_ndim - 2);
  while (sb.size() < pb.size()) sb.insert(sb.begin(), 1);
  while (pb.size() < sb.size()) pb.insert(pb.begin(), 1);
  const int nb = (int)sb.size();
  a->out_batch.resize(nb);
  for (int i = 0; i < nb; ++i) {
    if (sb[i] != pb[i] && sb[i] != 1 && pb[i] != 1) {
      *err = "Incompatible shapes: " + shape_str(d->source_shape, d->source_ndim) + " vs. " +
             shape_str(d->points_shape, d->points_ndim);
      return NUFFT_HIP_INVALID_ARGUMENT;
    }
    a->out_batch[i] = std::max(sb[i], pb[i]);
    if (sb[i] == 0 || pb[i] == 0) a->out_batch[i] = 0;
  }
  a->source_batch = sb;
  a->points_batch = pb;
  for (int i = 0; i < nb; ++i) {
    if (pb[i] == 1) {
      a->inner.push_back(i);
      a->num_transforms *= sb[i];
    } else {
      a->outer.push_back(i);
      a->num_calls *= pb[i];
    }
  }
  // transposition is needed iff some inner dim precedes an outer dim
  std::vector<int> perm(a->outer);
  perm.insert(perm.end(), a->inner.begin(), a->inner.end());
  for (int i = 0; i < nb; ++i)
    if (perm[i] != i) a->transpose = true;
  a->target_shape = a->out_batch;
  if (t1) a->target_shape.insert(a->target_shape.end(), a->grid.begin(), a->grid.end());
  else a->target_shape.push_back(a->num_points);
  return NUFFT_HIP_OK;
}

// ----------------------------------------------------------- plan cache

// Plans (and the batch-permute scratch buffers) are kept between calls: the reference
// rebuilds plan, FFT plan and tables on every Compute (nufft_kernels.cc:474-478). Least
// recently used first out, bounded by entry count and by device bytes.
struct CachedPlan {
  std::string key;
  nufft_hip_plan plan;
  hipStream_t own_stream;   // private stream of a pipelining lane (nullptr: caller's stream)
  int64_t bytes;
};
struct CachedScratch {      // permute temporaries of calls whose batch dims interleave
  void* stream;
  int device;
  void* ptr;
  size_t bytes;
};
std::mutex g_cache_mu;
std::list<CachedPlan> g_cache;         // most recently returned at the front
std::list<CachedScratch> g_scratch;
constexpr size_t kMaxCached = 16;
int64_t g_cache_limit = (int64_t)8 << 30;
constexpr int kMaxLanes = 4;   // pipelined plans per op call (nufft_hip_op_compute)

std::string plan_key(const nufft_hip_op_desc* d, const nufft_hip_options& opts, const Analysis& a, int type,
                     int ntransf, double tol, void* stream, int device, bool framework_alloc) {
  std::string k;
  char buf[256];
  snprintf(buf, sizeof(buf), "op%d t%d r%d n%d f%d p%d tol%.17g dev%d s%p a%d|", d->op_type, type, a.rank,
           ntransf, d->fft_direction, d->precision, tol, device, stream, framework_alloc ? 1 : 0);
  k = buf;
  for (auto g : a.grid) k += std::to_string((long long)g) + ",";
  k.append(reinterpret_cast<const char*>(&opts), sizeof(opts));
  return k;
}

nufft_hip_plan cache_take(const std::string& key, hipStream_t* own_stream) {
  std::lock_guard<std::mutex> lk(g_cache_mu);
  for (auto it = g_cache.begin(); it != g_cache.end(); ++it)
    if (it->key == key) {
      nufft_hip_plan p = it->plan;
      *own_stream = it->own_stream;
      g_cache.erase(it);
      return p;
    }
  return nullptr;
}

void release_entry(nufft_hip_plan p, hipStream_t own_stream) {
  nufft_hip_plan_destroy(p);   // synchronises the plan's stream
  if (own_stream) (void)hipStreamDestroy(own_stream);
}

int64_t cached_bytes_locked() {
  int64_t b = 0;
  for (auto& c : g_cache) b += c.bytes;
  for (auto& c : g_scratch) b += (int64_t)c.bytes;
  return b;
}

// Evicts from the cold end until both limits hold (the newest entry always stays).
void trim_cache() {
  for (;;) {
    CachedPlan evict{std::string(), nullptr, nullptr, 0};
    void* scratch = nullptr;
    {
      std::lock_guard<std::mutex> lk(g_cache_mu);
      const bool over = g_cache.size() > kMaxCached || cached_bytes_locked() > g_cache_limit;
      if (!over) return;
      if (!g_scratch.empty() && (g_cache.size() <= 1 || g_scratch.size() > 4)) {
        scratch = g_scratch.back().ptr;
        g_scratch.pop_back();
      } else if (g_cache.size() > 1) {
        evict = g_cache.back();
        g_cache.pop_back();
      } else if (!g_scratch.empty()) {
        scratch = g_scratch.back().ptr;
        g_scratch.pop_back();
      } else {
        return;
      }
    }
    if (scratch) (void)hipFree(scratch);   // hipFree waits for the device
    if (evict.plan) release_entry(evict.plan, evict.own_stream);
  }
}

void cache_give(const std::string& key, nufft_hip_plan p, hipStream_t own_stream) {
  nufft_hip_plan_info info;
  int64_t bytes = 0;
  if (nufft_hip_plan_get_info(p, &info) == NUFFT_HIP_OK) bytes = info.workspace_bytes;
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    g_cache.push_front({key, p, own_stream, bytes});
  }
  trim_cache();
}

// A scratch buffer of at least `bytes` for work enqueued on `stream`: reused only by
// later calls on the same stream, so stream order protects it. hipMalloc only on growth.
void* scratch_take(void* stream, int device, size_t bytes, size_t* got) {
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    for (auto it = g_scratch.begin(); it != g_scratch.end(); ++it)
      if (it->stream == stream && it->device == device && it->bytes >= bytes) {
        void* p = it->ptr;
        *got = it->bytes;
        g_scratch.erase(it);
        return p;
      }
  }
  void* p = nullptr;
  if (hipMalloc(&p, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  *got = bytes;
  return p;
}
void scratch_give(void* stream, int device, void* ptr, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_cache_mu);
  g_scratch.push_front({stream, device, ptr, bytes});
}

// ----------------------------------------------------------- options attr
// proto3 wire reader for the serialized `options` attr: the reference kernel calls
// Options::ParseFromString on it (nufft_kernels.cc:582-585; schema
// proto/nufft_options.proto:19-32). No generated code: four fields in three messages.
struct WireReader {
  const uint8_t* p;
  const uint8_t* end;
  bool varint(uint64_t* v) {   // at most 10 bytes, as the protobuf runtime accepts
    *v = 0;
    for (int i = 0; i < 10 && p < end; ++i) {
      const uint8_t b = *p++;
      *v |= (uint64_t)(b & 0x7F) << (7 * i);
      if (!(b & 0x80)) return true;
    }
    return false;
  }
  bool skip(size_t n) {
    if ((size_t)(end - p) < n) return false;
    p += n;
    return true;
  }
  // Skips the value of an unknown field (or of a known one sent with a foreign wire type,
  // which the runtime files under unknown fields as well). Groups nest.
  bool skip_value(uint32_t field, int wt, int depth) {
    uint64_t v;
    switch (wt) {
      case 0: return varint(&v);
      case 1: return skip(8);
      case 2: return varint(&v) && v <= (uint64_t)(end - p) && skip((size_t)v);
      case 5: return skip(4);
      case 3:   // start group: read fields until the matching end group
        if (depth > 64) return false;
        while (p < end) {
          uint64_t key;
          if (!varint(&key) || (key >> 3) == 0 || (key >> 3) > 0x1FFFFFFF) return false;
          if ((key & 7) == 4) return (uint32_t)(key >> 3) == field;
          if (!skip_value((uint32_t)(key >> 3), (int)(key & 7), depth + 1)) return false;
        }
        return false;
      default: return false;   // 4 (stray end group), 6, 7
    }
  }
};

// One message level: `on_field(field, wire_type, reader)` consumes the value of the
// fields it knows and returns 1, returns 0 for the ones it leaves to skip_value, -1 on error.
template <typename F>
bool parse_message(WireReader r, F&& on_field) {
  while (r.p < r.end) {
    uint64_t key;
    if (!r.varint(&key)) return false;
    const uint64_t field = key >> 3;
    const int wt = (int)(key & 7);
    if (field == 0 || field > 0x1FFFFFFF) return false;
    const int used = on_field((uint32_t)field, wt, &r);
    if (used < 0) return false;
    if (used == 0 && !r.skip_value((uint32_t)field, wt, 0)) return false;
  }
  return true;
}

// A length-delimited sub-message with one varint field number 1 (DebuggingOptions,
// FftwOptions). Repeated occurrences merge, the last value wins (proto3 semantics).
bool parse_single_varint_message(WireReader* r, uint64_t* value, bool* seen) {
  uint64_t len;
  if (!r->varint(&len) || len > (uint64_t)(r->end - r->p)) return false;
  WireReader sub{r->p, r->p + len};
  r->p += len;
  return parse_message(sub, [&](uint32_t field, int wt, WireReader* s) -> int {
    if (field != 1 || wt != 0) return 0;
    if (!s->varint(value)) return -1;
    *seen = true;
    return 1;
  });
}

}  // namespace

extern "C" {

int nufft_hip_options_from_proto(const void* bytes, size_t n, nufft_hip_options* out) {
  if (!out || (!bytes && n)) return NUFFT_HIP_INVALID_ARGUMENT;
  nufft_hip_options o;
  nufft_hip_default_options(&o);
  // proto3 leaves default-valued fields off the wire: an absent field 4 is
  // PointsRange.STRICT (= 0), which is what the parsed message hands the reference kernel
  // (nufft_kernels.cc:364-366). The Python wrapper's own default, EXTENDED (= 1), is
  // always on the wire (python/ops/nufft_options.py:222-273).
  o.points_range = NUFFT_HIP_RANGE_STRICT;
  o.max_batch_size = 0;
  o.check_points_range = 0;
  o.fftw_planning_rigor = 0;
  WireReader top{static_cast<const uint8_t*>(bytes), static_cast<const uint8_t*>(bytes) + n};
  const bool ok = parse_message(top, [&](uint32_t field, int wt, WireReader* r) -> int {
    uint64_t v = 0;
    bool seen = false;
    if (field == 1 && wt == 2) {          // DebuggingOptions debugging = 1 { bool check_points_range = 1 }
      if (!parse_single_varint_message(r, &v, &seen)) return -1;
      if (seen) o.check_points_range = v != 0;
      return 1;
    }
    if (field == 2 && wt == 2) {          // FftwOptions fftw = 2 { FftwPlanningRigor planning_rigor = 1 }
      if (!parse_single_varint_message(r, &v, &seen)) return -1;
      if (seen) o.fftw_planning_rigor = (int32_t)(uint32_t)v;
      return 1;
    }
    if (field == 3 && wt == 0) {          // int32 max_batch_size = 3 (low 32 bits of the varint)
      if (!r->varint(&v)) return -1;
      o.max_batch_size = (int32_t)(uint32_t)v;
      return 1;
    }
    if (field == 4 && wt == 0) {          // PointsRange points_range = 4 (open enum)
      if (!r->varint(&v)) return -1;
      o.points_range = (int32_t)(uint32_t)v;
      return 1;
    }
    return 0;
  });
  if (!ok) return NUFFT_HIP_INVALID_ARGUMENT;
  *out = o;
  return NUFFT_HIP_OK;
}

int nufft_hip_op_desc_from_attrs(nufft_hip_op_desc* desc, int op_type, const char* transform_type,
                                 const char* fft_direction, double tol, int precision,
                                 const void* options, size_t options_len,
                                 char* errbuf, size_t errbuf_len) {
  if (!desc) return fail(errbuf, errbuf_len, NUFFT_HIP_INVALID_ARGUMENT, "null desc");
  memset(desc, 0, sizeof(*desc));
  nufft_hip_default_options(&desc->options);
  desc->op_type = op_type;
  desc->precision = precision;
  desc->tol = tol;
  desc->fft_direction = NUFFT_HIP_FORWARD;
  if (precision != NUFFT_HIP_F32 && precision != NUFFT_HIP_F64)
    return fail(errbuf, errbuf_len, NUFFT_HIP_INVALID_ARGUMENT, "precision must be 4 (float) or 8 (double)");
  switch (op_type) {
    case NUFFT_HIP_OP_INTERP:   // nufft_kernels.cc:590-604
      desc->transform_type = NUFFT_HIP_TYPE_2;
      desc->fft_direction = NUFFT_HIP_BACKWARD;   // irrelevant, as upstream
      return NUFFT_HIP_OK;
    case NUFFT_HIP_OP_SPREAD:   // nufft_kernels.cc:607-621
      desc->transform_type = NUFFT_HIP_TYPE_1;
      return NUFFT_HIP_OK;
    case NUFFT_HIP_OP_NUFFT: break;
    default: return fail(errbuf, errbuf_len, NUFFT_HIP_INVALID_ARGUMENT, "unknown op type");
  }
  // nufft_kernels.cc:559-585. The op registration restricts both attrs to their two values
  // (nufft_ops.cc:211-212); anything else cannot reach the reference kernel and is refused here.
  const std::string tt = transform_type ? transform_type : "";
  const std::string fd = fft_direction ? fft_direction : "";
  if (tt == "type_1") desc->transform_type = NUFFT_HIP_TYPE_1;
  else if (tt == "type_2") desc->transform_type = NUFFT_HIP_TYPE_2;
  else return fail(errbuf, errbuf_len, NUFFT_HIP_INVALID_ARGUMENT,
                   "transform_type attr must be 'type_1' or 'type_2', but is " + tt);
  if (fd == "forward") desc->fft_direction = NUFFT_HIP_FORWARD;
  else if (fd == "backward") desc->fft_direction = NUFFT_HIP_BACKWARD;
  else return fail(errbuf, errbuf_len, NUFFT_HIP_INVALID_ARGUMENT,
                   "fft_direction attr must be 'forward' or 'backward', but is " + fd);
  if (nufft_hip_options_from_proto(options, options_len, &desc->options) != NUFFT_HIP_OK)
    return fail(errbuf, errbuf_len, NUFFT_HIP_INVALID_ARGUMENT, "Unable to parse options string.");
  return NUFFT_HIP_OK;
}

int nufft_hip_op_shape(const nufft_hip_op_desc* desc, int32_t* target_ndim, int64_t* target_shape,
                       char* errbuf, size_t errbuf_len) {
  if (!desc || !target_ndim || !target_shape)
    return fail(errbuf, errbuf_len, NUFFT_HIP_INVALID_ARGUMENT, "null argument");
  Analysis a;
  std::string err;
  int rc = analyze(desc, &a, &err);
  if (rc) return fail(errbuf, errbuf_len, rc, err);
  if (a.target_shape.size() > 12)
    return fail(errbuf, errbuf_len, NUFFT_HIP_INVALID_ARGUMENT, "too many dimensions");
  *target_ndim = (int32_t)a.target_shape.size();
  for (size_t i = 0; i < a.target_shape.size(); ++i) target_shape[i] = a.target_shape[i];
  return NUFFT_HIP_OK;
}

int nufft_hip_op_compute_ex(const nufft_hip_op_desc* desc, const void* source, const void* points,
                            void* target, void* stream_v, const nufft_hip_allocator* allocator,
                            char* errbuf, size_t errbuf_len) {
  if (!desc) return fail(errbuf, errbuf_len, NUFFT_HIP_INVALID_ARGUMENT, "null desc");
  Analysis a;
  std::string err;
  int rc = analyze(desc, &a, &err);
  if (rc) return fail(errbuf, errbuf_len, rc, err);
  hipStream_t stream = (hipStream_t)stream_v;
  const bool fw_alloc = allocator && allocator->alloc;
  const int rank = a.rank;
  const bool t1 = desc->transform_type == NUFFT_HIP_TYPE_1;
  const int nb = (int)a.source_batch.size();
  const size_t csize = 2 * (size_t)desc->precision;   // bytes per complex
  int64_t num_coeffs = 1;
  for (auto g : a.grid) num_coeffs *= g;
  int64_t out_elems = 1;
  for (auto s : a.target_shape) out_elems *= s;
  if (out_elems == 0) return NUFFT_HIP_OK;   // empty batch: nothing to do

  // grid dims reversed to x-fastest (nufft_kernels.cc:347-352)
  int64_t dims[3] = {1, 1, 1};
  for (int d = 0; d < rank; ++d) dims[d] = a.grid[rank - 1 - d];

  nufft_hip_options opts = desc->options;
  if (desc->op_type != NUFFT_HIP_OP_NUFFT) {   // nufft_kernels.cc:457-460
    opts.spread_only = 1;
    opts.upsampling_factor = 2.0;
  }
  const double tol = (double)(float)desc->tol;   // `tol: float` attr cast to FloatType (:361)
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess) {
    (void)hipGetLastError();
    return fail(errbuf, errbuf_len, NUFFT_HIP_INTERNAL,
                "no HIP device available (this library has no CPU fallback)");
  }
  // Calls (= distinct point sets of the batch, nufft_kernels.cc:491-540) are run in GROUPS of
  // up to K sets through one plan with K point sets (options.num_point_sets): one sort, one
  // spread, one batched FFT and one deconvolve launch for the whole group instead of ~12
  // launches per set -- a 512^2 / M = 1e6 item keeps the GPU busy for only ~60 us of the
  // ~90 us its own launches take. Needs the source to carry the same batch dims as the
  // points (otherwise the strengths of consecutive calls are not consecutive in memory).
  // options.op_group overrides K (1 = the r01 behaviour, one call at a time).
  // Groups alternate between two plans on private streams, so that the memory-bound sort of
  // one overlaps the LDS-bound spread of the other (r01: 0.126 -> 0.092 ms per item);
  // options.op_lanes overrides the lane count (1..kMaxLanes). With a framework allocator
  // everything stays on the caller's stream (its memory is ordered against that stream only).
  const int lanes_opt = desc->options.op_lanes, group_opt = desc->options.op_group;
  if (lanes_opt < 0 || group_opt < 0)
    return fail(errbuf, errbuf_len, NUFFT_HIP_INVALID_ARGUMENT, "options.op_group and options.op_lanes must be >= 0");
  std::vector<int64_t> src_outer, pts_outer;
  for (int i : a.outer) { src_outer.push_back(a.source_batch[i]); pts_outer.push_back(a.points_batch[i]); }
  bool groupable = a.num_calls > 1 && !a.transpose;
  for (size_t i = 0; i < src_outer.size(); ++i)
    if (src_outer[i] != pts_outer[i]) groupable = false;
  int64_t group = 1;
  if (groupable) {
    group = group_opt > 0 ? group_opt : 16;
    // (a plan takes at most 4096 point sets and 65535 transforms x sets: an oversized request is cut, not refused)
    group = std::min<int64_t>(group, std::min<int64_t>(4096, 65535 / std::max<int64_t>(1, a.num_transforms)));
    if (group < 1) group = 1;
    // (r06: the geometry of this configuration, memoised -- plan_describe builds a host plan (Gauss-Legendre rule, kernel fit):
    // ~150 us of host time on EVERY grouped call, more than the GPU needs for a group of 16 small transforms)
    nufft_hip_plan_info pi;
    bool described = false;
    {
      static std::mutex dmu;
      static std::map<std::string, nufft_hip_plan_info> dcache;
      const std::string dkey = plan_key(desc, opts, a, desc->transform_type, (int)a.num_transforms, tol, nullptr, 0, false);
      std::lock_guard<std::mutex> lk(dmu);
      const auto it = dcache.find(dkey);
      if (it != dcache.end()) {
        pi = it->second;
        described = true;
      } else {
        char pe[256];
        described = nufft_hip_plan_describe(desc->transform_type, rank, dims, desc->fft_direction, (int)a.num_transforms, tol,
                                            desc->precision, &opts, &pi, pe, sizeof(pe)) == NUFFT_HIP_OK;
        if (described) {
          if (dcache.size() >= 256) dcache.clear();
          dcache[dkey] = pi;
        }
      }
    }
    if (described) {
      const double fine_bytes = (double)pi.fine_dims[0] * pi.fine_dims[1] * pi.fine_dims[2] * csize *
                                (opts.spread_only ? 0.0 : (double)a.num_transforms);
      const int64_t ntiles = (int64_t)pi.num_tiles[0] * pi.num_tiles[1] * pi.num_tiles[2];
      // (slots = transforms x sets ride in grid.y of the batched FFT / deconvolve launches: <= 65535)
      while (group > 1 && (fine_bytes * group > 1.5 * (1 << 30) || ntiles * group > 65536 ||
                           a.num_points * group > 1500000000LL || a.num_transforms * group > 32768))
        group /= 2;
    } else {
      group = 1;
    }
    group = std::min<int64_t>(group, a.num_calls);
    if (group < 2) group = 1;
  }
  const int64_t ngroups = (a.num_calls + group - 1) / group;
  const int64_t tail = a.num_calls - (ngroups - 1) * group;   // sets in the last group
  int want_lanes = 2;
  if (lanes_opt > 0) want_lanes = lanes_opt < kMaxLanes ? lanes_opt : kMaxLanes;
  if (fw_alloc) want_lanes = 1;
  const int64_t nfull = tail != group ? ngroups - 1 : ngroups;   // groups of the full size (>= 1)
  const int nlanes = (int)std::min<int64_t>(nfull, want_lanes);
  // plans: one per lane for full groups, plus one for a shorter last group
  const int nplans = nlanes + (tail != group ? 1 : 0);
  nufft_hip_plan plans[kMaxLanes + 1] = {};
  hipStream_t lane_stream[kMaxLanes + 1] = {};
  std::string keys[kMaxLanes + 1];
  for (int l = 0; l < nplans; ++l) {
    const bool is_tail = l == nlanes;   // the shorter last group runs on the caller's stream, after the lanes joined
    nufft_hip_options o = opts;
    o.num_point_sets = (int32_t)(is_tail ? tail : group);
    if (o.num_point_sets <= 1) o.num_point_sets = 0;
    const bool own_lane = nlanes > 1 && !is_tail;
    void* key_stream = own_lane ? reinterpret_cast<void*>((intptr_t)(l + 1)) : stream_v;
    keys[l] = plan_key(desc, o, a, desc->transform_type, (int)a.num_transforms, tol, key_stream, device, fw_alloc);
    plans[l] = cache_take(keys[l], &lane_stream[l]);
    if (plans[l] && fw_alloc) rc = nufft_hip_plan_set_allocator(plans[l], allocator);   // this call's context
    if (!plans[l]) {
      if (own_lane && hipStreamCreateWithFlags(&lane_stream[l], hipStreamNonBlocking) != hipSuccess) {
        for (int k = 0; k < l; ++k) release_entry(plans[k], lane_stream[k]);
        return fail(errbuf, errbuf_len, NUFFT_HIP_INTERNAL, "hipStreamCreate failed");
      }
      char pe[512] = {0};
      rc = nufft_hip_plan_create_ex(&plans[l], desc->transform_type, rank, dims, desc->fft_direction,
                                    (int)a.num_transforms, tol, desc->precision, &o,
                                    own_lane ? (void*)lane_stream[l] : stream_v, allocator, pe, sizeof(pe));
      if (rc) {
        if (lane_stream[l]) (void)hipStreamDestroy(lane_stream[l]);
        for (int k = 0; k < l; ++k) release_entry(plans[k], lane_stream[k]);
        return fail(errbuf, errbuf_len, rc, pe);
      }
    }
  }
  nufft_hip_plan plan = plans[0];
  auto release_all = [&]() {
    for (int l = 0; l < nplans; ++l) release_entry(plans[l], lane_stream[l]);
  };

  // Source / target with batch dims permuted to [outer..., inner..., element...]
  // when the original order interleaves them (nufft_kernels.cc:241-345,372-378).
  // The temporaries come from the framework allocator, or from a scratch buffer cached
  // per stream (no hipMalloc / hipFree / synchronisation on the steady-state path).
  const void* psource = source;
  void* ptarget = target;
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  std::vector<int> perm(a.outer);
  perm.insert(perm.end(), a.inner.begin(), a.inner.end());
  const int s_nd = nb + a.source_elem_rank;
  const int t_nd = (int)a.target_shape.size();
  std::vector<int64_t> sshape(a.source_batch);
  for (int i = desc->source_ndim - a.source_elem_rank; i < desc->source_ndim; ++i) sshape.push_back(desc->source_shape[i]);
  auto contiguous_strides = [](const std::vector<int64_t>& s) {
    std::vector<int64_t> st(s.size(), 1);
    for (int i = (int)s.size() - 2; i >= 0; --i) st[i] = st[i + 1] * s[i + 1];
    return st;
  };
  auto cleanup = [&]() {
    if (!scratch) return;
    if (fw_alloc) {
      if (allocator->free) allocator->free(scratch, allocator->user);
    } else {
      scratch_give(stream_v, device, scratch, scratch_bytes);
    }
    scratch = nullptr;
  };
  auto hip_fail = [&](hipError_t e) {
    cleanup();
    release_all();
    return fail(errbuf, errbuf_len, NUFFT_HIP_INTERNAL, std::string("HIP error: ") + hipGetErrorString(e));
  };
  if (a.transpose) {
    // tsource[outer.., inner.., elem..] = source[perm]
    std::vector<int64_t> sst = contiguous_strides(sshape), oshape(s_nd), ostr(s_nd);
    for (int i = 0; i < s_nd; ++i) {
      const int src_dim = i < nb ? perm[i] : i;
      oshape[i] = sshape[src_dim];
      ostr[i] = sst[src_dim];
    }
    int64_t n = 1;
    for (auto v : oshape) n *= v;
    const size_t sbytes = (std::max<size_t>(16, (size_t)n * csize) + 255) & ~(size_t)255;
    const size_t tbytes = std::max<size_t>(16, (size_t)out_elems * csize);
    if (fw_alloc) {
      scratch = allocator->alloc(sbytes + tbytes, allocator->user);
      scratch_bytes = sbytes + tbytes;
    } else {
      scratch = scratch_take(stream_v, device, sbytes + tbytes, &scratch_bytes);
    }
    if (!scratch) {
      release_all();
      return fail(errbuf, errbuf_len, NUFFT_HIP_RESOURCE_EXHAUSTED, "out of device memory for the batch-permute temporaries");
    }
    hipError_t e = launch_permute(source, scratch, (int)csize, s_nd, oshape.data(), ostr.data(), stream);
    if (e != hipSuccess) return hip_fail(e);
    psource = scratch;
    ptarget = (char*)scratch + sbytes;
  }

  // Loop over calls (nufft_kernels.cc:491-540), `group` of them per plan call. Batch dims in
  // `outer` order.
  const int no = (int)a.outer.size();
  std::vector<int64_t> sfac(no, 1), pfac(no, 1);
  for (int d2 = no - 2; d2 >= 0; --d2) {
    sfac[d2] = sfac[d2 + 1] * src_outer[d2 + 1];
    pfac[d2] = pfac[d2 + 1] * pts_outer[d2 + 1];
  }
  const size_t rsize = (size_t)desc->precision;
  // (r06: the fork / join events live in a per-thread, per-device pool -- creating and destroying three of them was part
  // of every multi-lane call's host time)
  auto pooled_event = [&](int i, hipEvent_t* ev) {
    static thread_local std::map<int, std::vector<hipEvent_t>> pool;
    std::vector<hipEvent_t>& v = pool[device];
    if ((int)v.size() <= i) v.resize(i + 1, nullptr);
    if (!v[i]) {
      const hipError_t e = hipEventCreateWithFlags(&v[i], hipEventDisableTiming);
      if (e != hipSuccess) { v[i] = nullptr; return e; }
    }
    *ev = v[i];
    return hipSuccess;
  };
  hipEvent_t ev_fork = nullptr, ev_join[kMaxLanes] = {};
  if (nlanes > 1) {
    hipError_t e = pooled_event(0, &ev_fork);
    if (e == hipSuccess) e = hipEventRecord(ev_fork, stream);
    for (int l = 0; l < nlanes && e == hipSuccess; ++l) e = hipStreamWaitEvent(lane_stream[l], ev_fork, 0);
    if (e != hipSuccess) return hip_fail(e);
  }
  auto run_group = [&](int64_t gi, nufft_hip_plan pl) {
    const int64_t call = gi * group;
    plan = pl;
    const char* pb = (const char*)points + (size_t)call * (size_t)a.num_points * rank * rsize;
    // x = LAST coordinate of each point (reverse of the last axis, :282-286)
    const void* px = pb + (size_t)(rank - 1) * rsize;
    const void* py = rank > 1 ? pb + (size_t)(rank - 2) * rsize : nullptr;
    const void* pz = rank > 2 ? pb + (size_t)(rank - 3) * rsize : nullptr;
    int64_t source_index = 0, tmp = call;
    for (int d2 = 0; d2 < no; ++d2) {
      int64_t ix = tmp / pfac[d2];
      tmp %= pfac[d2];
      if (src_outer[d2] == 1) ix = 0;
      source_index += ix * sfac[d2];
    }
    const int64_t target_index = call;
    const int64_t c_index = t1 ? source_index : target_index;
    const int64_t f_index = t1 ? target_index : source_index;
    char* cbase = (char*)(t1 ? const_cast<void*>(psource) : ptarget);
    char* fbase = (char*)(t1 ? ptarget : const_cast<void*>(psource));
    void* c = cbase + (size_t)c_index * (size_t)a.num_transforms * (size_t)a.num_points * csize;
    void* f = fbase + (size_t)f_index * (size_t)a.num_transforms * (size_t)num_coeffs * csize;
    // set_points + execute / interp / spread as one call (:492-539): the plan may fuse them.
    // A grouped plan takes its K point sets, strength blocks and grids from consecutive memory.
    return nufft_hip_execute_with_points(pl, a.num_points, px, py, pz, rank, c, f);
  };
  for (int64_t gi = 0; gi < nfull && !rc; ++gi) rc = run_group(gi, plans[gi % nlanes]);
  if (rc) {
    const std::string msg = nufft_hip_last_error(plan);
    cleanup();
    release_all();
    return fail(errbuf, errbuf_len, rc, msg);
  }
  if (nlanes > 1) {   // join: the caller's stream continues after both lanes
    hipError_t e = hipSuccess;
    for (int l = 0; l < nlanes && e == hipSuccess; ++l) {
      e = pooled_event(1 + l, &ev_join[l]);
      if (e == hipSuccess) e = hipEventRecord(ev_join[l], lane_stream[l]);
      if (e == hipSuccess) e = hipStreamWaitEvent(stream, ev_join[l], 0);
    }
    if (e != hipSuccess) return hip_fail(e);
  }
  if (tail != group) {   // the shorter last group, on the caller's stream
    rc = run_group(ngroups - 1, plans[nlanes]);
    if (rc) {
      const std::string msg = nufft_hip_last_error(plans[nlanes]);
      cleanup();
      release_all();
      return fail(errbuf, errbuf_len, rc, msg);
    }
  }
  if (a.transpose) {
    // target[original order] = ttarget[outer.., inner.., elem..]: for output dim j
    // (original position), its stride in ttarget is that of position iperm[j].
    std::vector<int64_t> tshape_perm(t_nd);
    for (int i = 0; i < t_nd; ++i) tshape_perm[i] = a.target_shape[i < nb ? perm[i] : i];
    std::vector<int64_t> tst = contiguous_strides(tshape_perm), ostr(t_nd);
    std::vector<int> iperm(t_nd);
    for (int i = 0; i < t_nd; ++i) iperm[i < nb ? perm[i] : i] = i;
    for (int j = 0; j < t_nd; ++j) ostr[j] = tst[iperm[j]];
    hipError_t e = launch_permute(ptarget, target, (int)csize, t_nd, a.target_shape.data(), ostr.data(), stream);
    if (e != hipSuccess) return hip_fail(e);
  }
  cleanup();
  for (int l = 0; l < nplans; ++l) {
    // a framework allocator's memory is per call: hand the workspace back, keep the plan
    if (fw_alloc) (void)nufft_hip_plan_release_workspace(plans[l]);
    cache_give(keys[l], plans[l], lane_stream[l]);
  }
  return NUFFT_HIP_OK;
}

int nufft_hip_op_compute(const nufft_hip_op_desc* desc, const void* source, const void* points,
                         void* target, void* stream_v, char* errbuf, size_t errbuf_len) {
  return nufft_hip_op_compute_ex(desc, source, points, target, stream_v, nullptr, errbuf, errbuf_len);
}

void nufft_hip_op_clear_cache(void) {
  std::list<CachedPlan> tmp;
  std::list<CachedScratch> tmp2;
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    tmp.swap(g_cache);
    tmp2.swap(g_scratch);
  }
  for (auto& c : tmp) release_entry(c.plan, c.own_stream);
  for (auto& c : tmp2) (void)hipFree(c.ptr);
}

void nufft_hip_op_set_cache_limit(int64_t max_bytes) {
  {
    std::lock_guard<std::mutex> lk(g_cache_mu);
    g_cache_limit = max_bytes < 0 ? 0 : max_bytes;
  }
  trim_cache();
}

int64_t nufft_hip_op_cache_bytes(void) {
  std::lock_guard<std::mutex> lk(g_cache_mu);
  return cached_bytes_locked();
}

}  // extern "C"
