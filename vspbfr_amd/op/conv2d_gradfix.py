"""conv2d_gradfix: signature mirror of reference op/conv2d_gradfix.py:22-92 (which, on every torch other than
1.7/1.8, is a pass-through to F.conv2d / F.conv_transpose2d, :78-92) routed to the gfx950 implicit-GEMM kernel.
Forward only.  `groups > 1` (the reference's `groups=batch` modulated form, models/RestoreNet.py:373-383) is ONE launch: the kernel's
true-group mode (x_group_stride) gives every group its own input-channel slice; the model code in this package does not need it
(it uses the modulate-input / demodulate-output form, see vspbfr_amd/layers.py)."""
import contextlib

import torch

from .. import hip_ops

enabled = True
weight_gradients_disabled = False


@contextlib.contextmanager
def no_weight_gradients():
    global weight_gradients_disabled
    old = weight_gradients_disabled
    weight_gradients_disabled = True
    yield
    weight_gradients_disabled = old


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    (sy, sx), (py, px), (dy, dx) = _pair(stride), _pair(padding), _pair(dilation)
    if sy != sx or py != px or dy != dx:
        raise RuntimeError("conv2d_gradfix.conv2d: only square stride/padding/dilation are supported")
    if groups == 1:
        return hip_ops.conv2d(input.contiguous(), weight.contiguous(), bias, sy, py, dy)
    B, C, H, W = input.shape
    cout, cg, kh, kw = weight.shape
    if C != cg * groups or cout % groups:
        raise RuntimeError(f"conv2d_gradfix.conv2d: input has {C} channels, weight expects {cg} x {groups} groups")
    pc = hip_ops.PackedConv(hip_ops.pack_weight(weight.contiguous(), groups), groups, cout // groups, cg, kh, kw, sy, (dy,), (py,),
                            x_group_stride=cg)
    return hip_ops.conv2d_packed(input.contiguous(), pc, ch_bias=None if bias is None else bias.contiguous())


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    (sy, sx) = _pair(stride)
    if (sy, sx) != (2, 2) or _pair(padding) != (0, 0) or _pair(output_padding) != (0, 0) or _pair(dilation) != (1, 1) \
            or tuple(weight.shape[2:]) != (3, 3) or bias is not None:
        raise RuntimeError("conv2d_gradfix.conv_transpose2d: the restoration path only uses stride=2, padding=0, 3x3, "
                           "no bias (models/RestoreNet.py:530-532); other forms are not implemented")
    B, C, H, W = input.shape
    cg = C // groups
    outs = []
    for g in range(groups):
        w_io = weight[g * cg:(g + 1) * cg]  # (Cin_g, Cout_g, 3, 3)
        phases = hip_ops.pack_transposed_s2(w_io.transpose(0, 1).contiguous())
        outs.append(hip_ops.conv_transpose2d_s2(input[:, g * cg:(g + 1) * cg].contiguous(), phases))
    return outs[0] if groups == 1 else torch.cat(outs, dim=1)
