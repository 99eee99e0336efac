#include "binding_common.h"

// upfirdn2d.upfirdn2d(input[major, H, W, minor], kernel[kh, kw], up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
// reference op/upfirdn2d.cpp:17-31
torch::Tensor upfirdn2d(const torch::Tensor& input, const torch::Tensor& kernel, int up_x, int up_y, int down_x, int down_y, int pad_x0,
                        int pad_x1, int pad_y0, int pad_y1) {
  VSP_CHECK_INPUT(input);
  VSP_CHECK_INPUT(kernel);
  VSP_CHECK_SAME_DEVICE(kernel, input);
  VSP_DEVICE_GUARD(input);   // reference op/upfirdn2d.cpp:23
  TORCH_CHECK(input.dim() == 4 && kernel.dim() == 2, "upfirdn2d expects input [major,H,W,minor] and a 2-D kernel");
  const int major = (int)input.size(0), in_h = (int)input.size(1), in_w = (int)input.size(2), minor = (int)input.size(3);
  const int kh = (int)kernel.size(0), kw = (int)kernel.size(1);
  const int out_h = (in_h * up_y + pad_y0 + pad_y1 - kh + down_y) / down_y, out_w = (in_w * up_x + pad_x0 + pad_x1 - kw + down_x) / down_x;
  TORCH_CHECK(out_h >= 0 && out_w >= 0, "upfirdn2d: negative output size");
  TORCH_CHECK(kernel.scalar_type() == input.scalar_type(), "kernel must have the input's dtype");
  const auto x = vsp_f32(input), k32 = vsp_f32(kernel);
  auto out = at::empty({major, out_h, out_w, minor}, x.options());
  vsp_raise(vsp_upfirdn2d_f32(out.data_ptr<float>(), x.data_ptr<float>(), k32.data_ptr<float>(), major, in_h, in_w, minor, kh, kw, up_x,
                              up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, nullptr, vsp_current_stream()),
            "upfirdn2d");
  return input.scalar_type() == at::kFloat ? out : out.to(input.scalar_type());
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) { m.def("upfirdn2d", &upfirdn2d, "upfirdn2d (gfx950)"); }
