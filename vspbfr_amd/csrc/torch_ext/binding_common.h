// pybind11 torch-extension front ends of the C ABI (include/vspbfr_hip.h): the two native modules the reference JIT-builds at import,
// `fused` (op/fused_act.py:13-20 -> op/fused_bias_act.cpp:18-31) and `upfirdn2d` (op/upfirdn2d.py:13-20 -> op/upfirdn2d.cpp:17-31),
// with the reference's exact function names and argument lists, so that its op/*.py binds to the gfx950 kernels by replacing the
// `load(...)` call with `import fused` / `import upfirdn2d`.  No kernels live here: tensor checks,
// output allocation (what the reference does with torch::empty_like / at::empty inside the op) and the call on the current stream.
#pragma once
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include "../../../include/vspbfr_hip.h"

// Element types: the reference's modules dispatch over float, double and half (AT_DISPATCH_FLOATING_TYPES_AND_HALF,
// op/fused_bias_act_kernel.cu:96, op/upfirdn2d_kernel.cu:311).  The gfx950 kernels compute in fp32; half and double tensors are
// converted at this boundary (one cast launch each way) and the result is returned in the input's type -- for half that is what the
// reference's kernel does internally (it accumulates in the tensor's scalar type, i.e. LESS precisely), for double the arithmetic is
// fp32 (documented deviation: the restoration path is fp32 end to end).
#define VSP_CHECK_INPUT(x)                                                                                           \
  TORCH_CHECK((x).is_cuda(), #x " must be a CUDA tensor");                                                           \
  TORCH_CHECK((x).is_contiguous(), #x " must be contiguous");                                                        \
  TORCH_CHECK((x).scalar_type() == at::kFloat || (x).scalar_type() == at::kHalf || (x).scalar_type() == at::kDouble, \
              #x " must be float32, float16 or float64")

inline torch::Tensor vsp_f32(const torch::Tensor& x) { return x.scalar_type() == at::kFloat ? x : x.to(at::kFloat); }

// Device guard on the input's device, as the reference's front ends do (op/fused_bias_act.cpp:25, op/upfirdn2d.cpp:23
// `const at::cuda::OptionalCUDAGuard device_guard(device_of(input))`): the launch goes to the TENSOR's device and that device's current
// stream, whatever the caller's current device is; the previous device is restored on return.  Operands on another device are refused.
#define VSP_DEVICE_GUARD(x) const c10::hip::OptionalHIPGuardMasqueradingAsCUDA vsp_device_guard(at::device_of(x))
#define VSP_CHECK_SAME_DEVICE(x, ref) TORCH_CHECK((x).device() == (ref).device(), #x " is on ", (x).device(), ", expected ", (ref).device())

inline void* vsp_current_stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }
inline void vsp_raise(int rc, const char* what) { TORCH_CHECK(rc == 0, what, ": ", vsp_last_error()); }
