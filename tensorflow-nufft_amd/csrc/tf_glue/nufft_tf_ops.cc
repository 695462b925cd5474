// TensorFlow (ROCm build) custom-op glue over libnufft_hip.so.
//
// Re-registers the reference's ops `NUFFT`, `Interp`, `Spread` with the same
// names, inputs, attrs and shape functions (reference
// tensorflow_nufft/cc/ops/nufft_ops.cc:27-219) and GPU kernels with
// HostMemory("grid_shape") (reference cc/kernels/nufft_kernels.cc:624-706).
// All host logic lives behind the C ABI (include/nufft_hip.h): attr decoding incl. the
// options proto (nufft_hip_op_desc_from_attrs / nufft_hip_options_from_proto), validation,
// batch broadcasting and the call loop (nufft_hip_op_shape / nufft_hip_op_compute_ex); all
// of it is tested through ctypes (tests/test_cabi_cpu.py, tests/test_options.py). This file
// only moves attrs, shapes and device pointers between TF and those calls. The one piece
// of logic that must live here is the graph-time shape function (TF's InferenceContext
// works on symbolic shapes).
//
// NOT BUILT IN THIS REPOSITORY'S IMAGE: TensorFlow is not installed here. What IS
// checked here: this file type-checks (g++ -fsyntax-only) against tests/tf_api_stub/, a set of declarations
// with the names and signatures of TensorFlow's public C++ op API and no behaviour
// (tests/test_cabi_cpu.py::test_tf_glue_type_checks_against_the_api_stub) -- syntax, types and the use of the
// C ABI, nothing about the real headers or about loading. The build line for a TensorFlow-ROCm machine is in
// INTEGRATION.md section 1.
#define EIGEN_USE_GPU
#include <string>
#include <vector>

#include "nufft_hip.h"
#include "tensorflow/core/framework/common_shape_fns.h"
#include "tensorflow/core/framework/op.h"
#include "tensorflow/core/framework/op_kernel.h"
#include "tensorflow/core/framework/shape_inference.h"

namespace tensorflow {
namespace nufft_hip_glue {

using shape_inference::DimensionHandle;
using shape_inference::InferenceContext;
using shape_inference::ShapeHandle;

// Shape function: same checks and messages as reference NUFFTBaseShapeFn
// (nufft_ops.cc:27-103).
Status BaseShapeFn(InferenceContext* c, int transform_type) {
  ShapeHandle source_shape = c->input(0);
  ShapeHandle points_shape = c->input(1);
  DimensionHandle unused;
  DimensionHandle rank_handle = c->Dim(points_shape, -1);
  if (!(c->WithValue(rank_handle, 1, &unused).ok() || c->WithValue(rank_handle, 2, &unused).ok() ||
        c->WithValue(rank_handle, 3, &unused).ok())) {
    return errors::InvalidArgument("Dimension must be 1, 2 or 3, but is ", c->DebugString(rank_handle));
  }
  if (!c->ValueKnown(rank_handle)) {
    c->set_output(0, c->UnknownShape());
    return OkStatus();
  }
  const int64_t rank = c->Value(rank_handle);
  ShapeHandle grid_shape;
  if (transform_type == 1) {
    TF_RETURN_IF_ERROR(c->MakeShapeFromShapeTensor(2, &grid_shape));
    TF_RETURN_IF_ERROR(c->WithRank(grid_shape, rank, &grid_shape));
  }
  DimensionHandle num_points = c->Dim(points_shape, -2);
  if (transform_type == 1) TF_RETURN_IF_ERROR(c->Merge(num_points, c->Dim(source_shape, -1), &num_points));
  const int64_t first_elem_axis = transform_type == 1 ? -1 : -rank;
  ShapeHandle source_batch, points_batch, out_batch, out;
  TF_RETURN_IF_ERROR(c->Subshape(source_shape, 0, first_elem_axis, &source_batch));
  TF_RETURN_IF_ERROR(c->Subshape(points_shape, 0, -2, &points_batch));
  TF_RETURN_IF_ERROR(shape_inference::BroadcastBinaryOpOutputShapeFnHelper(c, source_batch, points_batch,
                                                                          true, &out_batch));
  if (transform_type == 1) TF_RETURN_IF_ERROR(c->Concatenate(out_batch, grid_shape, &out));
  else TF_RETURN_IF_ERROR(c->Concatenate(out_batch, c->Vector(num_points), &out));
  c->set_output(0, out);
  return OkStatus();
}

Status NUFFTShapeFn(InferenceContext* c) {
  string t;
  TF_RETURN_IF_ERROR(c->GetAttr("transform_type", &t));
  if (t == "type_1") return BaseShapeFn(c, 1);
  if (t == "type_2") return BaseShapeFn(c, 2);
  return errors::InvalidArgument("transform_type attr must be 'type_1' or 'type_2', but is ", t);
}

REGISTER_OP("Interp")
    .Attr("Tcomplex: {complex64, complex128} = DT_COMPLEX64")
    .Attr("Treal: {float32, float64} = DT_FLOAT")
    .Input("source: Tcomplex").Input("points: Treal").Output("target: Tcomplex")
    .Attr("tol: float = 1e-6")
    .SetShapeFn([](InferenceContext* c) { return BaseShapeFn(c, 2); });
REGISTER_OP("Spread")
    .Attr("Tcomplex: {complex64, complex128} = DT_COMPLEX64")
    .Attr("Treal: {float32, float64} = DT_FLOAT")
    .Attr("Tshape: {int32, int64} = DT_INT32")
    .Input("source: Tcomplex").Input("points: Treal").Input("grid_shape: Tshape").Output("target: Tcomplex")
    .Attr("tol: float = 1e-6")
    .SetShapeFn([](InferenceContext* c) { return BaseShapeFn(c, 1); });
REGISTER_OP("NUFFT")
    .Attr("Tcomplex: {complex64, complex128} = DT_COMPLEX64")
    .Attr("Treal: {float32, float64} = DT_FLOAT")
    .Attr("Tshape: {int32, int64} = DT_INT32")
    .Input("source: Tcomplex").Input("points: Treal").Input("grid_shape: Tshape").Output("target: Tcomplex")
    .Attr("transform_type: {'type_1', 'type_2'} = 'type_2'")
    .Attr("fft_direction: {'forward', 'backward'} = 'forward'")
    .Attr("tol: float = 1e-6")
    .Attr("options: string = ''")
    .SetShapeFn(NUFFTShapeFn);

template <typename FloatType>
class NufftHipOp : public OpKernel {
 public:
  // Attrs -> descriptor: nufft_hip_op_desc_from_attrs restates the reference constructors
  // (nufft_kernels.cc:559-621), including Options::ParseFromString on the `options` attr.
  NufftHipOp(OpKernelConstruction* ctx, int op_type) : OpKernel(ctx) {
    float tol;
    string transform_type, fft_direction, options;
    OP_REQUIRES_OK(ctx, ctx->GetAttr("tol", &tol));
    if (op_type == NUFFT_HIP_OP_NUFFT) {
      OP_REQUIRES_OK(ctx, ctx->GetAttr("transform_type", &transform_type));
      OP_REQUIRES_OK(ctx, ctx->GetAttr("fft_direction", &fft_direction));
      OP_REQUIRES_OK(ctx, ctx->GetAttr("options", &options));
    }
    char err[256] = {0};
    const int rc = nufft_hip_op_desc_from_attrs(&desc_, op_type, transform_type.c_str(), fft_direction.c_str(),
                                                tol, sizeof(FloatType), options.data(), options.size(),
                                                err, sizeof(err));
    OP_REQUIRES(ctx, rc == NUFFT_HIP_OK, ToStatus(rc, err));
  }

  void Compute(OpKernelContext* ctx) override {
    const Tensor& source = ctx->input(0);
    const Tensor& points = ctx->input(1);
    nufft_hip_op_desc d = desc_;   // attrs; shapes below
    OP_REQUIRES(ctx, source.dims() <= 12 && points.dims() <= 12, errors::InvalidArgument("too many dimensions"));
    d.source_ndim = source.dims();
    d.points_ndim = points.dims();
    for (int i = 0; i < source.dims(); ++i) d.source_shape[i] = source.dim_size(i);
    for (int i = 0; i < points.dims(); ++i) d.points_shape[i] = points.dim_size(i);
    if (d.transform_type == NUFFT_HIP_TYPE_1) {
      const Tensor& gs = ctx->input(2);   // host memory
      OP_REQUIRES(ctx, TensorShapeUtils::IsVector(gs.shape()),
                  errors::InvalidArgument("grid_shape must be 1D, but got shape: ", gs.shape().DebugString()));
      d.grid_shape_len = static_cast<int32_t>(gs.dim_size(0));
      for (int i = 0; i < gs.dim_size(0) && i < 3; ++i)
        d.grid_shape[i] = gs.dtype() == DT_INT32 ? gs.vec<int32>()(i) : gs.vec<int64_t>()(i);
    }
    char err[1024] = {0};
    int32_t ndim = 0;
    int64_t shape[12];
    int rc = nufft_hip_op_shape(&d, &ndim, shape, err, sizeof(err));
    OP_REQUIRES(ctx, rc == NUFFT_HIP_OK, ToStatus(rc, err));
    TensorShape target_shape;
    for (int i = 0; i < ndim; ++i) target_shape.AddDim(shape[i]);
    Tensor* target = nullptr;
    OP_REQUIRES_OK(ctx, ctx->allocate_output(0, target_shape, &target));
    // TF-ROCm exposes the compute stream as a hipStream_t through the Eigen device.
    void* stream = reinterpret_cast<void*>(ctx->eigen_gpu_device().stream());
    // workspace and batch-permute temporaries come from allocate_temp, like the reference
    // plan's fine grid (nufft_plan.cu.cc:1981-1986); the tensors die with this Compute
    struct TempPool { OpKernelContext* ctx; std::vector<Tensor> keep; } pool{ctx, {}};
    nufft_hip_allocator alloc;
    alloc.user = &pool;
    alloc.alloc = [](size_t bytes, void* user) -> void* {
      TempPool* p = static_cast<TempPool*>(user);
      Tensor t;
      if (!p->ctx->allocate_temp(DT_UINT8, TensorShape({static_cast<int64_t>(bytes)}), &t).ok()) return nullptr;
      p->keep.push_back(t);
      return t.data();
    };
    alloc.free = [](void*, void*) {};   // released when `pool` goes out of scope (stream-ordered by TF)
    rc = nufft_hip_op_compute_ex(&d, source.data(), points.data(), target->data(), stream, &alloc, err, sizeof(err));
    OP_REQUIRES(ctx, rc == NUFFT_HIP_OK, ToStatus(rc, err));
  }

 private:
  static Status ToStatus(int rc, const char* msg) {
    switch (rc) {
      case NUFFT_HIP_INVALID_ARGUMENT: return errors::InvalidArgument(msg);
      case NUFFT_HIP_UNIMPLEMENTED: return errors::Unimplemented(msg);
      case NUFFT_HIP_RESOURCE_EXHAUSTED: return errors::ResourceExhausted(msg);
      default: return errors::Internal(msg);
    }
  }
  nufft_hip_op_desc desc_;
};

template <typename F> struct NUFFT : NufftHipOp<F> { explicit NUFFT(OpKernelConstruction* c) : NufftHipOp<F>(c, NUFFT_HIP_OP_NUFFT) {} };
template <typename F> struct Interp : NufftHipOp<F> { explicit Interp(OpKernelConstruction* c) : NufftHipOp<F>(c, NUFFT_HIP_OP_INTERP) {} };
template <typename F> struct Spread : NufftHipOp<F> { explicit Spread(OpKernelConstruction* c) : NufftHipOp<F>(c, NUFFT_HIP_OP_SPREAD) {} };

#define REGISTER_GPU(NAME, F, C)                                                             \
  REGISTER_KERNEL_BUILDER(Name(#NAME).Device(DEVICE_GPU).TypeConstraint<C>("Tcomplex")       \
                              .TypeConstraint<F>("Treal").HostMemory("grid_shape"), NAME<F>)
REGISTER_GPU(NUFFT, float, complex64);
REGISTER_GPU(NUFFT, double, complex128);
REGISTER_GPU(Spread, float, complex64);
REGISTER_GPU(Spread, double, complex128);
REGISTER_KERNEL_BUILDER(Name("Interp").Device(DEVICE_GPU).TypeConstraint<complex64>("Tcomplex").TypeConstraint<float>("Treal"), Interp<float>);
REGISTER_KERNEL_BUILDER(Name("Interp").Device(DEVICE_GPU).TypeConstraint<complex128>("Tcomplex").TypeConstraint<double>("Treal"), Interp<double>);

}  // namespace nufft_hip_glue
}  // namespace tensorflow
