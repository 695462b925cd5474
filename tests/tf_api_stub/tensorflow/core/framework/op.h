// API stand-in for tests only -- see tests/tf_api_stub/README.md. NOT TensorFlow.
#ifndef TF_API_STUB_OP_H_
#define TF_API_STUB_OP_H_
#include <complex>
#include <cstdint>
#include <map>
#include <sstream>
#include <string>
#include <vector>
namespace tensorflow {
using string = std::string;
using int32 = int32_t;
using complex64 = std::complex<float>;
using complex128 = std::complex<double>;
enum DataType { DT_INT32 = 3, DT_UINT8 = 4, DT_INT64 = 9 };
// Status: a code and a message (enough for the glue's error paths to be observable)
class Status {
 public:
  Status() = default;
  Status(int code, std::string msg) : code_(code), msg_(std::move(msg)) {}
  bool ok() const { return code_ == 0; }
  int code() const { return code_; }
  const std::string& message() const { return msg_; }
 private:
  int code_ = 0;
  std::string msg_;
};
inline Status OkStatus() { return Status(); }
namespace errors {
template <typename... A> std::string StubStrCat(const A&... a) {
  std::ostringstream os;
  ((os << a), ...);
  return os.str();
}
template <typename... A> Status InvalidArgument(const A&... a) { return Status(3, StubStrCat(a...)); }
template <typename... A> Status Unimplemented(const A&... a) { return Status(12, StubStrCat(a...)); }
template <typename... A> Status ResourceExhausted(const A&... a) { return Status(8, StubStrCat(a...)); }
template <typename... A> Status Internal(const A&... a) { return Status(13, StubStrCat(a...)); }
}  // namespace errors
#define TF_RETURN_IF_ERROR(...)                  \
  do {                                           \
    ::tensorflow::Status _s = (__VA_ARGS__);     \
    if (!_s.ok()) return _s;                     \
  } while (0)
namespace shape_inference { class InferenceContext; }
using StubShapeFn = Status (*)(shape_inference::InferenceContext*);
// REGISTER_OP keeps the op's name and shape function in a table the test driver looks them up in
inline std::map<std::string, StubShapeFn>& StubOpRegistry() {
  static std::map<std::string, StubShapeFn> r;
  return r;
}
class OpDefBuilderWrapper {
 public:
  explicit OpDefBuilderWrapper(const char* name) : name_(name) {}
  OpDefBuilderWrapper& Attr(const std::string&) { return *this; }
  OpDefBuilderWrapper& Input(const std::string&) { return *this; }
  OpDefBuilderWrapper& Output(const std::string&) { return *this; }
  OpDefBuilderWrapper& SetShapeFn(StubShapeFn fn) {
    StubOpRegistry()[name_] = fn;
    return *this;
  }
 private:
  std::string name_;
};
#define TF_STUB_CAT2(a, b) a##b
#define TF_STUB_CAT(a, b) TF_STUB_CAT2(a, b)
#define REGISTER_OP(name) static ::tensorflow::OpDefBuilderWrapper TF_STUB_CAT(tf_stub_op_, __COUNTER__) = ::tensorflow::OpDefBuilderWrapper(name)
}  // namespace tensorflow
#endif
