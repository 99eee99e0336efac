#include "binding_common.h"

// fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale): reference op/fused_bias_act.cpp:18-31
torch::Tensor fused_bias_act(const torch::Tensor& input, const torch::Tensor& bias, const torch::Tensor& refer, int act, int grad,
                             float alpha, float scale) {
  VSP_CHECK_INPUT(input);
  if (bias.numel()) { VSP_CHECK_INPUT(bias); VSP_CHECK_SAME_DEVICE(bias, input); }
  if (refer.numel()) { VSP_CHECK_INPUT(refer); VSP_CHECK_SAME_DEVICE(refer, input); }
  VSP_DEVICE_GUARD(input);   // reference op/fused_bias_act.cpp:25
  TORCH_CHECK(!bias.numel() || bias.scalar_type() == input.scalar_type(), "bias must have the input's dtype");
  TORCH_CHECK(!refer.numel() || refer.scalar_type() == input.scalar_type(), "refer must have the input's dtype");
  const auto x = vsp_f32(input.contiguous());
  const auto bias32 = bias.numel() ? vsp_f32(bias) : bias;
  const auto refer32 = refer.numel() ? vsp_f32(refer) : refer;
  auto y = torch::empty_like(x);
  int64_t step_b = 1;
  for (int i = 2; i < x.dim(); ++i) step_b *= x.size(i);
  TORCH_CHECK(!refer.numel() || refer.numel() == x.numel(), "refer must have the same number of elements as input");
  vsp_raise(vsp_fused_bias_act_f32(y.data_ptr<float>(), x.data_ptr<float>(), bias.numel() ? bias32.data_ptr<float>() : nullptr,
                                   refer.numel() ? refer32.data_ptr<float>() : nullptr, x.numel(), (int)step_b, (int)bias.numel(), act, grad,
                                   alpha, scale, vsp_current_stream()),
            "fused_bias_act");
  return input.scalar_type() == at::kFloat ? y : y.to(input.scalar_type());
}

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) { m.def("fused_bias_act", &fused_bias_act, "fused bias act (gfx950)"); }
