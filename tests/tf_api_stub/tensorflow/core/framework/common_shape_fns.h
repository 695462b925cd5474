// API stand-in for tests only -- see tests/tf_api_stub/README.md. NOT TensorFlow.
#ifndef TF_API_STUB_COMMON_SHAPE_FNS_H_
#define TF_API_STUB_COMMON_SHAPE_FNS_H_
#include <algorithm>
#include "tensorflow/core/framework/shape_inference.h"
namespace tensorflow {
namespace shape_inference {
// numpy-style broadcast of two shapes with unknowns, as TensorFlow's helper of the same name resolves them
// (common_shape_fns.cc): unknown rank -> unknown shape; per dimension pair (right-aligned): equal or one of them 1;
// an unknown paired with a known > 1 takes that value, with 1 or another unknown stays unknown.
inline Status BroadcastBinaryOpOutputShapeFnHelper(InferenceContext* c, ShapeHandle x, ShapeHandle y,
                                                   bool incompatible_shape_error, ShapeHandle* out) {
  if (!x.known_rank || !y.known_rank) { *out = c->UnknownShape(); return OkStatus(); }
  const size_t rx = x.dims.size(), ry = y.dims.size(), r = std::max(rx, ry);
  std::vector<int64_t> dims(r);
  for (size_t i = 0; i < r; ++i) {
    const int64_t a = i < r - rx ? 1 : x.dims[i - (r - rx)];
    const int64_t b = i < r - ry ? 1 : y.dims[i - (r - ry)];
    if (a < 0 || b < 0) {
      if (a > 1) dims[i] = a;
      else if (b > 1) dims[i] = b;
      else dims[i] = -1;
    } else if (a == 1 || b == 1) {
      dims[i] = a == 1 ? b : a;
    } else if (a == b) {
      dims[i] = a;
    } else {
      if (!incompatible_shape_error) { *out = c->UnknownShape(); return OkStatus(); }
      *out = {};
      return errors::InvalidArgument("Dimensions must be equal, but are ", a, " and ", b);
    }
  }
  *out = {true, dims};
  return OkStatus();
}
}  // namespace shape_inference
}  // namespace tensorflow
#endif
