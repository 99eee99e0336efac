"""fused bias + leaky ReLU: API mirror of reference op/fused_act.py:199-233 over the gfx950 kernel.

`fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)` has the reference's native-module signature
(op/fused_bias_act.cpp:18-31).  The restoration path runs under torch.no_grad(); with autograd enabled the op is
differentiable to any order like the reference's (op/fused_act.py:126-196): y = lrelu(x + b) * scale has the piecewise
constant derivative m(y) = scale * (y > 0 ? 1 : slope), so every derivative is the SAME kernel call
`fused_bias_act(g, -, refer=y, act=3, grad=1)` = g * m(y), which is linear in g."""
import torch
from torch import nn
from torch.autograd import Function

from .. import hip_ops


class _FusedModule:
    """Stand-in for the pybind11 module `fused` that the reference JIT-builds at import (op/fused_act.py:13-20)."""

    @staticmethod
    def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
        return hip_ops.fused_bias_act(input, bias, refer, act, grad, alpha, scale)


fused = _FusedModule()


class _SlopeMask(Function):
    """g -> g * m(y) (+ optional per-channel bias added to g first, the form the double backward needs)."""

    @staticmethod
    def forward(ctx, g, y, gbias, negative_slope, scale):
        ctx.save_for_backward(y)
        ctx.cfg = (negative_slope, scale, gbias is not None)
        empty = g.new_empty(0)
        return fused.fused_bias_act(g.contiguous(), gbias if gbias is not None else empty, y, 3, 1, negative_slope, scale)

    @staticmethod
    def backward(ctx, gg):
        (y,) = ctx.saved_tensors
        slope, scale, has_bias = ctx.cfg
        d = _SlopeMask.apply(gg, y, None, slope, scale)          # linear in g: the adjoint is the same mask
        return d, None, (_channel_sum(d) if has_bias else None), None, None


def _channel_sum(t):
    if torch.is_grad_enabled() or t.dtype != torch.float32:     # (a double-backward pass differentiates through the sum)
        return t.sum([0] + list(range(2, t.ndim)))
    return hip_ops.channel_sum(t.contiguous())


class _LeakyBias(Function):
    @staticmethod
    def forward(ctx, x, bias, negative_slope, scale):
        empty = x.new_empty(0)
        y = fused.fused_bias_act(x.contiguous(), bias if bias is not None else empty, empty, 3, 0, negative_slope, scale)
        ctx.save_for_backward(y)
        ctx.cfg = (negative_slope, scale, bias is not None)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        slope, scale, has_bias = ctx.cfg
        gx = _SlopeMask.apply(g, y, None, slope, scale)
        return gx, (_channel_sum(gx) if has_bias else None), None, None


def fused_leaky_relu(input, bias=None, negative_slope=0.2, scale=2 ** 0.5):
    if input.device.type != "cuda":
        raise RuntimeError("vspbfr_amd.op.fused_leaky_relu: input must be a CUDA (HIP) tensor; the CPU statement of "
                           "this op lives in oracle/ and is test-only")
    if torch.is_grad_enabled() and (input.requires_grad or (bias is not None and bias.requires_grad)):
        return _LeakyBias.apply(input, bias, negative_slope, scale)
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
