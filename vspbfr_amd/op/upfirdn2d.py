"""upfirdn2d: API mirror of reference op/upfirdn2d.py:346-362 over the gfx950 kernel.

Differentiable to any order with respect to the input (reference op/upfirdn2d.py:217-343): the adjoint of
"zero-insert by `up`, pad, correlate with the flipped taps, keep every `down`-th sample" is again an upfirdn2d -- taps
flipped, `up` and `down` exchanged, and the padding that maps the output grid back onto the input grid -- so the backward
pass is this same Function applied to the gradient, recursively."""
from collections import abc

import torch
from torch.autograd import Function

from .. import hip_ops


class _UpfirdnModule:
    """Stand-in for the pybind11 module `upfirdn2d` of the reference (op/upfirdn2d.py:13-20, op/upfirdn2d.cpp:17-31)."""

    @staticmethod
    def upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
        return hip_ops.upfirdn2d_native_layout(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)


upfirdn2d_op = _UpfirdnModule()


def _run(x, kernel, up, down, pad):
    B, C, h, w = x.shape
    out = upfirdn2d_op.upfirdn2d(x.reshape(-1, h, w, 1).contiguous(), kernel.contiguous(), up[0], up[1], down[0], down[1],
                                 pad[0], pad[1], pad[2], pad[3])
    return out.view(B, C, out.shape[1], out.shape[2])


class _Fir(Function):
    @staticmethod
    def forward(ctx, x, kernel, up, down, pad):
        y = _run(x, kernel, up, down, pad)
        kh, kw = kernel.shape
        in_h, in_w = x.shape[2:]
        out_h, out_w = y.shape[2:]
        # padding of the adjoint: the first gradient tap that can reach input sample 0, and whatever is left at the far
        # end so that the adjoint's output has exactly the input's size
        adj_pad = (kw - pad[0] - 1, in_w * up[0] - out_w * down[0] + pad[0] - up[0] + 1,
                   kh - pad[2] - 1, in_h * up[1] - out_h * down[1] + pad[2] - up[1] + 1)
        ctx.save_for_backward(kernel)
        ctx.adjoint = (down, up, adj_pad)
        return y

    @staticmethod
    def backward(ctx, g):
        (kernel,) = ctx.saved_tensors
        a_up, a_down, a_pad = ctx.adjoint
        gx = _Fir.apply(g, torch.flip(kernel, [0, 1]), a_up, a_down, a_pad) if ctx.needs_input_grad[0] else None
        return gx, None, None, None, None


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    if not isinstance(up, abc.Iterable):
        up = (up, up)
    if not isinstance(down, abc.Iterable):
        down = (down, down)
    if len(pad) == 2:
        pad = (pad[0], pad[1], pad[0], pad[1])
    if input.device.type != "cuda":
        raise RuntimeError("vspbfr_amd.op.upfirdn2d: input must be a CUDA (HIP) tensor; the CPU statement of this op "
                           "lives in oracle/ and is test-only")
    if torch.is_grad_enabled() and input.requires_grad:
        return _Fir.apply(input, kernel, tuple(up), tuple(down), tuple(pad))
    return _run(input, kernel, up, down, pad)
