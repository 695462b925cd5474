// API stand-in for tests only -- see tests/tf_api_stub/README.md. NOT TensorFlow.
// The dozen InferenceContext calls the glue's shape function makes, with the behaviour TensorFlow documents for
// them (tensorflow/core/framework/shape_inference.h): shapes of known or unknown rank whose dimensions are known
// or unknown (-1), negative indices counted from the end, Merge / WithValue / WithRank refining unknowns and
// refusing contradictions with TensorFlow's messages.
#ifndef TF_API_STUB_SHAPE_INFERENCE_H_
#define TF_API_STUB_SHAPE_INFERENCE_H_
#include <limits>
#include "tensorflow/core/framework/op.h"
namespace tensorflow {
namespace shape_inference {
struct DimensionHandle {
  int64_t v = -1;   // -1: unknown
};
struct ShapeHandle {
  bool known_rank = false;
  std::vector<int64_t> dims;   // -1: unknown dimension
};
class InferenceContext {
 public:
  static constexpr int64_t kUnknownDim = -1;
  // test set-up
  std::vector<ShapeHandle> inputs;
  std::map<int, std::vector<int64_t>> input_values;      // constant value of a shape-tensor input (entries may be -1)
  std::map<std::string, std::string> string_attrs;
  std::vector<ShapeHandle> outputs = std::vector<ShapeHandle>(1);

  ShapeHandle input(int64_t i) { return inputs.at(i); }
  bool RankKnown(const ShapeHandle& s) { return s.known_rank; }
  int64_t Rank(const ShapeHandle& s) { return s.known_rank ? (int64_t)s.dims.size() : -1; }
  DimensionHandle Dim(const ShapeHandle& s, int64_t idx) {
    if (!s.known_rank) return {};
    if (idx < 0) idx += (int64_t)s.dims.size();
    return {s.dims.at(idx)};
  }
  bool ValueKnown(DimensionHandle d) { return d.v >= 0; }
  int64_t Value(DimensionHandle d) { return d.v; }
  std::string DebugString(DimensionHandle d) { return d.v >= 0 ? std::to_string(d.v) : "?"; }
  std::string DebugString(const ShapeHandle& s) {
    if (!s.known_rank) return "?";
    std::string r = "[";
    for (size_t i = 0; i < s.dims.size(); ++i) r += (i ? "," : "") + DebugString(DimensionHandle{s.dims[i]});
    return r + "]";
  }
  Status WithValue(DimensionHandle d, int64_t value, DimensionHandle* out) {
    if (d.v < 0) { *out = {value}; return OkStatus(); }
    if (d.v == value) { *out = d; return OkStatus(); }
    *out = {};
    return errors::InvalidArgument("Dimension must be ", value, " but is ", d.v);
  }
  void set_output(int i, ShapeHandle s) { outputs.at(i) = s; }
  ShapeHandle UnknownShape() { return {}; }
  ShapeHandle UnknownShapeOfRank(int64_t rank) { return {true, std::vector<int64_t>((size_t)rank, -1)}; }
  ShapeHandle Vector(DimensionHandle d) { return {true, {d.v}}; }
  Status MakeShapeFromShapeTensor(int input_idx, ShapeHandle* out) {
    const ShapeHandle t = input(input_idx);
    if (t.known_rank && t.dims.size() != 1) return errors::InvalidArgument("Shape must be rank 1 but is rank ", t.dims.size());
    const auto it = input_values.find(input_idx);
    if (it != input_values.end()) {
      for (int64_t v : it->second)
        if (v < -1) return errors::InvalidArgument("Invalid value in tensor used for shape: ", v);
      *out = {true, it->second};
      return OkStatus();
    }
    const DimensionHandle n = Dim(t, 0);   // length of the shape vector = rank of the shape
    *out = n.v >= 0 ? UnknownShapeOfRank(n.v) : UnknownShape();
    return OkStatus();
  }
  Status WithRank(const ShapeHandle& s, int64_t rank, ShapeHandle* out) {
    if (!s.known_rank) { *out = UnknownShapeOfRank(rank); return OkStatus(); }
    if ((int64_t)s.dims.size() == rank) { *out = s; return OkStatus(); }
    const Status bad = errors::InvalidArgument("Shape must be rank ", rank, " but is rank ", s.dims.size());
    *out = {};   // (may alias s)
    return bad;
  }
  Status Merge(DimensionHandle a, DimensionHandle b, DimensionHandle* out) {
    if (a.v < 0) { *out = b; return OkStatus(); }
    if (b.v < 0 || a.v == b.v) { *out = a; return OkStatus(); }
    *out = {};
    return errors::InvalidArgument("Dimensions must be equal, but are ", a.v, " and ", b.v);
  }
  Status Subshape(const ShapeHandle& s, int64_t start, int64_t end, ShapeHandle* out) {
    const int64_t kmax = std::numeric_limits<int64_t>::max();
    if (start == 0 && end == kmax) { *out = s; return OkStatus(); }
    if (!s.known_rank) { *out = UnknownShape(); return OkStatus(); }
    const int64_t rank = (int64_t)s.dims.size();
    if (start > rank) start = rank;
    if (end > rank) end = rank;
    if (start < 0) {
      start += rank;
      if (start < 0) { *out = {}; return errors::InvalidArgument("Subshape start out of bounds: ", start - rank, ", for shape with rank ", rank); }
    }
    if (end < 0) {
      end += rank;
      if (end < 0) { *out = {}; return errors::InvalidArgument("Subshape end out of bounds: ", end - rank, ", for shape with rank ", rank); }
    }
    if (start > end) { *out = {}; return errors::InvalidArgument("Subshape must have computed start <= end, but is ", start, " and ", end); }
    *out = {true, std::vector<int64_t>(s.dims.begin() + start, s.dims.begin() + end)};
    return OkStatus();
  }
  Status Concatenate(const ShapeHandle& a, const ShapeHandle& b, ShapeHandle* out) {
    if (!a.known_rank || !b.known_rank) { *out = UnknownShape(); return OkStatus(); }
    *out = a;
    out->dims.insert(out->dims.end(), b.dims.begin(), b.dims.end());
    return OkStatus();
  }
  template <typename T> Status GetAttr(const std::string& name, T* value) {
    const auto it = string_attrs.find(name);
    if (it == string_attrs.end()) return errors::InvalidArgument("No attr named '", name, "' in NodeDef");
    *value = it->second;
    return OkStatus();
  }
};
}  // namespace shape_inference
}  // namespace tensorflow
#endif
