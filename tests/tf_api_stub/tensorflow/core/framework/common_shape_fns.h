// API stub for type-checking only -- see tests/tf_api_stub/README.md. NOT TensorFlow.
#ifndef TF_API_STUB_COMMON_SHAPE_FNS_H_
#define TF_API_STUB_COMMON_SHAPE_FNS_H_
#include "tensorflow/core/framework/shape_inference.h"
namespace tensorflow {
namespace shape_inference {
inline Status BroadcastBinaryOpOutputShapeFnHelper(InferenceContext*, ShapeHandle, ShapeHandle, bool, ShapeHandle*) { return {}; }
}  // namespace shape_inference
}  // namespace tensorflow
#endif
