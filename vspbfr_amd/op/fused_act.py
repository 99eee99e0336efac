"""fused bias + leaky ReLU: API mirror of reference op/fused_act.py:199-233 over the gfx950 kernel.

`fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)` has the reference's native-module signature
(op/fused_bias_act.cpp:18-31).  Forward only: the path under construction is inference (torch.no_grad)."""
import torch
from torch import nn

from .. import hip_ops


class _FusedModule:
    """Stand-in for the pybind11 module `fused` that the reference JIT-builds at import (op/fused_act.py:13-20)."""

    @staticmethod
    def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
        return hip_ops.fused_bias_act(input, bias, refer, act, grad, alpha, scale)


fused = _FusedModule()


def fused_leaky_relu(input, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    if input.device.type != "cuda":
        raise RuntimeError("vspbfr_amd.op.fused_leaky_relu: input must be a CUDA (HIP) tensor; the CPU statement of "
                           "this op lives in oracle/ and is test-only")
    if input.requires_grad and torch.is_grad_enabled():
        raise RuntimeError("vspbfr_amd.op.fused_leaky_relu is forward-only (inference path); wrap in torch.no_grad()")
    empty = input.new_empty(0)
    return fused.fused_bias_act(input.contiguous(), bias if bias is not None else empty, empty, 3, 0, negative_slope,
                                scale)


class FusedLeakyReLU(nn.Module):
    def __init__(self, channel, bias=True, negative_slope=0.2, scale=2 ** 0.5):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel)) if bias else None
        self.negative_slope = negative_slope
        self.scale = scale

    def forward(self, input):
        return fused_leaky_relu(input, self.bias, self.negative_slope, self.scale)
