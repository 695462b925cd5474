// API stub for type-checking only -- see tests/tf_api_stub/README.md. NOT TensorFlow.
#ifndef TF_API_STUB_OP_H_
#define TF_API_STUB_OP_H_
#include <complex>
#include <cstdint>
#include <string>
#include <vector>
namespace tensorflow {
using string = std::string;
using int32 = int32_t;
using complex64 = std::complex<float>;
using complex128 = std::complex<double>;
enum DataType { DT_INT32 = 3, DT_UINT8 = 4, DT_INT64 = 9 };
class Status {
 public:
  Status() = default;
  bool ok() const { return true; }
};
inline Status OkStatus() { return Status(); }
namespace errors {
template <typename... A> Status InvalidArgument(A...) { return Status(); }
template <typename... A> Status Unimplemented(A...) { return Status(); }
template <typename... A> Status ResourceExhausted(A...) { return Status(); }
template <typename... A> Status Internal(A...) { return Status(); }
}  // namespace errors
#define TF_RETURN_IF_ERROR(...)                  \
  do {                                           \
    ::tensorflow::Status _s = (__VA_ARGS__);     \
    if (!_s.ok()) return _s;                     \
  } while (0)
namespace shape_inference { class InferenceContext; }
class OpDefBuilderWrapper {
 public:
  explicit OpDefBuilderWrapper(const char*) {}
  OpDefBuilderWrapper& Attr(const std::string&) { return *this; }
  OpDefBuilderWrapper& Input(const std::string&) { return *this; }
  OpDefBuilderWrapper& Output(const std::string&) { return *this; }
  OpDefBuilderWrapper& SetShapeFn(Status (*)(shape_inference::InferenceContext*)) { return *this; }
};
#define TF_STUB_CAT2(a, b) a##b
#define TF_STUB_CAT(a, b) TF_STUB_CAT2(a, b)
#define REGISTER_OP(name) static ::tensorflow::OpDefBuilderWrapper TF_STUB_CAT(tf_stub_op_, __COUNTER__) = ::tensorflow::OpDefBuilderWrapper(name)
}  // namespace tensorflow
#endif
