// pybind11 torch-extension front ends of the C ABI (include/vspbfr_hip.h): the two native modules the reference JIT-builds at import,
// `fused` (op/fused_act.py:13-20 -> op/fused_bias_act.cpp:18-31) and `upfirdn2d` (op/upfirdn2d.py:13-20 -> op/upfirdn2d.cpp:17-31),
// with the reference's exact function names and argument lists, so that its op/*.py binds to the gfx950 kernels by replacing the
// `load(...)` call with `import fused` / `import upfirdn2d`.  No kernels live here: tensor checks,
// output allocation (what the reference does with torch::empty_like / at::empty inside the op) and the call on the current stream.
#pragma once
#include <torch/extension.h>
#include <c10/hip/HIPStream.h>
#include "../../../include/vspbfr_hip.h"

#define VSP_CHECK_INPUT(x)                                                         \
  TORCH_CHECK((x).is_cuda(), #x " must be a CUDA tensor");                         \
  TORCH_CHECK((x).is_contiguous(), #x " must be contiguous");                      \
  TORCH_CHECK((x).scalar_type() == at::kFloat, #x " must be float32")

inline void* vsp_current_stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }
inline void vsp_raise(int rc, const char* what) { TORCH_CHECK(rc == 0, what, ": ", vsp_last_error()); }
