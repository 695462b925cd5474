// API stub for type-checking only -- see tests/tf_api_stub/README.md. NOT TensorFlow.
#ifndef TF_API_STUB_OP_KERNEL_H_
#define TF_API_STUB_OP_KERNEL_H_
#include <cstring>
#include <initializer_list>
#include "tensorflow/core/framework/op.h"
namespace tensorflow {
class TensorShape {
 public:
  TensorShape() = default;
  TensorShape(std::initializer_list<int64_t>) {}
  void AddDim(int64_t) {}
  std::string DebugString() const { return {}; }
};
struct TensorShapeUtils { static bool IsVector(const TensorShape&) { return true; } };
template <typename T> struct TfStubVec { T operator()(int64_t) const { return T(); } };
class Tensor {
 public:
  int dims() const { return 0; }
  int64_t dim_size(int) const { return 0; }
  void* data() const { return nullptr; }
  const TensorShape& shape() const { return shape_; }
  DataType dtype() const { return DT_INT32; }
  template <typename T> TfStubVec<T> vec() const { return {}; }
 private:
  TensorShape shape_;
};
struct TfStubGpuDevice { void* stream() const { return nullptr; } };   // Eigen::GpuDevice::stream() is a hipStream_t on TF-ROCm
class OpKernelConstruction {
 public:
  template <typename T> Status GetAttr(const std::string&, T*) const { return {}; }
  void CtxFailure(const Status&) {}
};
class OpKernelContext {
 public:
  const Tensor& input(int) { return t_; }
  Status allocate_output(int, const TensorShape&, Tensor**) { return {}; }
  Status allocate_temp(DataType, const TensorShape&, Tensor*) { return {}; }
  const TfStubGpuDevice& eigen_gpu_device() const { return d_; }
  void CtxFailure(const Status&) {}
 private:
  Tensor t_;
  TfStubGpuDevice d_;
};
class OpKernel {
 public:
  explicit OpKernel(OpKernelConstruction*) {}
  virtual ~OpKernel() = default;
  virtual void Compute(OpKernelContext*) = 0;
};
#define OP_REQUIRES(CTX, EXP, STATUS)  \
  do {                                 \
    if (!(EXP)) {                      \
      (CTX)->CtxFailure((STATUS));     \
      return;                          \
    }                                  \
  } while (0)
#define OP_REQUIRES_OK(CTX, ...)                  \
  do {                                            \
    ::tensorflow::Status _s(__VA_ARGS__);         \
    if (!_s.ok()) {                               \
      (CTX)->CtxFailure(_s);                      \
      return;                                     \
    }                                             \
  } while (0)
constexpr const char* DEVICE_GPU = "GPU";
class KernelDefBuilder {
 public:
  explicit KernelDefBuilder(const char*) {}
  KernelDefBuilder& Device(const char*) { return *this; }
  template <typename T> KernelDefBuilder& TypeConstraint(const char*) { return *this; }
  KernelDefBuilder& HostMemory(const char*) { return *this; }
};
inline KernelDefBuilder Name(const char* n) { return KernelDefBuilder(n); }
// instantiates the kernel class (so that its member functions are type-checked) without registering anything
#define REGISTER_KERNEL_BUILDER(BUILDER, ...)                                                             \
  static ::tensorflow::OpKernel* TF_STUB_CAT(tf_stub_make_, __COUNTER__)(::tensorflow::OpKernelConstruction* c) { \
    (void)(BUILDER);                                                                                      \
    return new __VA_ARGS__(c);                                                                            \
  }
}  // namespace tensorflow
#endif
