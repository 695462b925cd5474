// API stub for type-checking only -- see tests/tf_api_stub/README.md. NOT TensorFlow.
#ifndef TF_API_STUB_SHAPE_INFERENCE_H_
#define TF_API_STUB_SHAPE_INFERENCE_H_
#include "tensorflow/core/framework/op.h"
namespace tensorflow {
namespace shape_inference {
class DimensionHandle {};
class ShapeHandle {};
class InferenceContext {
 public:
  ShapeHandle input(int64_t) { return {}; }
  DimensionHandle Dim(ShapeHandle, int64_t) { return {}; }
  Status WithValue(DimensionHandle, int64_t, DimensionHandle*) { return {}; }
  bool ValueKnown(DimensionHandle) { return true; }
  int64_t Value(DimensionHandle) { return 0; }
  void set_output(int, ShapeHandle) {}
  ShapeHandle UnknownShape() { return {}; }
  Status MakeShapeFromShapeTensor(int, ShapeHandle*) { return {}; }
  Status WithRank(ShapeHandle, int64_t, ShapeHandle*) { return {}; }
  Status Merge(DimensionHandle, DimensionHandle, DimensionHandle*) { return {}; }
  Status Subshape(ShapeHandle, int64_t, int64_t, ShapeHandle*) { return {}; }
  Status Concatenate(ShapeHandle, ShapeHandle, ShapeHandle*) { return {}; }
  ShapeHandle Vector(DimensionHandle) { return {}; }
  std::string DebugString(DimensionHandle) { return {}; }
  template <typename T> Status GetAttr(const std::string&, T*) { return {}; }
};
}  // namespace shape_inference
}  // namespace tensorflow
#endif
