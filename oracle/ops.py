"""CPU statements of the path's operators (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Each function cites the reference definition it restates.  fp32 on CPU tensors throughout.
"""
import math

import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)


def fused_bias_act(x, bias, ref, act, grad, alpha, scale):
    """Native op semantics (reference op/fused_bias_act_kernel.cu:27-63): bias broadcast over dim 1, then
    act*10+grad selects identity (10, 11), leaky-relu (30), ref-gated leaky-relu (31) or zero (12, 32); times scale."""
    v = x
    if bias is not None and bias.numel():
        v = v + bias.view(1, -1, *([1] * (x.dim() - 2)))
    mode = act * 10 + grad
    if mode == 30:
        y = torch.where(v > 0, v, v * alpha)
    elif mode == 31:
        y = torch.where(ref > 0, v, v * alpha)
    elif mode in (12, 32):
        y = torch.zeros_like(v)
    else:
        y = v
    return y * scale


def fused_leaky_relu(x, bias=None, negative_slope=0.2, scale=SQRT2):
    """reference op/fused_act.py:216-233 (the CPU branch hard-codes slope 0.2, :222,228; the CUDA branch passes the
    argument through -- identical for the only value the path uses)."""
    return fused_bias_act(x, bias, None, 3, 0, negative_slope, scale)


def upfirdn2d(x, kernel, up=1, down=1, pad=(0, 0)):
    """NCHW statement of reference op/upfirdn2d.py:346-406: zero-insert by `up`, pad (negative = crop), convolve with
    the kernel (true convolution = correlation with the flipped taps, :392), keep every `down`-th sample."""
    up_x, up_y = (up, up) if isinstance(up, int) else up
    down_x, down_y = (down, down) if isinstance(down, int) else down
    if len(pad) == 2:
        px0, px1, py0, py1 = pad[0], pad[1], pad[0], pad[1]
    else:
        px0, px1, py0, py1 = pad
    B, Cc, H, W = x.shape
    kh, kw = kernel.shape
    z = x.new_zeros(B, Cc, H * up_y, W * up_x)
    z[:, :, ::up_y, ::up_x] = x
    z = F.pad(z, [max(px0, 0), max(px1, 0), max(py0, 0), max(py1, 0)])
    z = z[:, :, max(-py0, 0): z.shape[2] - max(-py1, 0), max(-px0, 0): z.shape[3] - max(-px1, 0)]
    w = torch.flip(kernel, [0, 1]).view(1, 1, kh, kw)
    out = F.conv2d(z.reshape(B * Cc, 1, z.shape[2], z.shape[3]), w)
    out = out[:, :, ::down_y, ::down_x]
    return out.reshape(B, Cc, out.shape[2], out.shape[3])


def make_kernel(k):
    """reference models/RestoreNet.py:32-40."""
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    return k / k.sum()


def pixel_norm(x):
    """reference models/RestoreNet.py:28-29 / models/CodeDiffuser.py:11-12: normalise over dim 1."""
    return x * torch.rsqrt(torch.mean(x ** 2, dim=1, keepdim=True) + 1e-8)


def equal_linear(x, weight, bias, lr_mul=1.0, activation=False):
    """reference models/RestoreNet.py:161-171 (identical copy e4e/models/stylegan2/model.py:151-160)."""
    scale = (1 / math.sqrt(weight.shape[1])) * lr_mul
    if activation:
        return fused_leaky_relu(F.linear(x, weight * scale), bias * lr_mul)
    return F.linear(x, weight * scale, bias=bias * lr_mul)


def modulated_conv(x, weight, style, demodulate=True, mode="same", dilation=1, blur_kernel=None):
    """reference models/RestoreNet.py:510-555 / :373-418 (fused branch): per-sample weight = scale * W * style,
    demodulated over (ci, ky, kx), applied as a grouped conv with groups = batch.
    weight: (1, Cout, Cin, k, k); style: (B, Cin) already passed through the modulation EqualLinear.
    mode: "same" (stride 1, padding = dilation*(k-1)//2), "up" (conv_transpose2d stride 2 then blur pad (1,1),
    :443-449,530-535), "down" (blur pad (2,2) then stride-2 conv, :451-457,538-545).  `blur_kernel` is the module's
    registered buffer (already x4 for "up", models/RestoreNet.py:91-92)."""
    B, Cin, H, W = x.shape
    _, Cout, _, k, _ = weight.shape
    scale = 1 / math.sqrt(Cin * k * k)
    w = scale * weight * style.view(B, 1, Cin, 1, 1)
    if demodulate:
        d = torch.rsqrt(w.pow(2).sum([2, 3, 4]) + 1e-8)
        w = w * d.view(B, Cout, 1, 1, 1)
    if mode == "up":
        xin = x.reshape(1, B * Cin, H, W)
        wt = w.transpose(1, 2).reshape(B * Cin, Cout, k, k)
        out = F.conv_transpose2d(xin, wt, padding=0, stride=2, groups=B)
        out = out.view(B, Cout, out.shape[2], out.shape[3])
        return upfirdn2d(out, blur_kernel, pad=(1, 1))
    w = w.view(B * Cout, Cin, k, k)
    if mode == "down":
        xb = upfirdn2d(x, blur_kernel, pad=(2, 2))
        out = F.conv2d(xb.reshape(1, B * Cin, xb.shape[2], xb.shape[3]), w, padding=0, stride=2, groups=B)
    else:
        pad = ((k - 1) * dilation) // 2
        out = F.conv2d(x.reshape(1, B * Cin, H, W), w, padding=pad, groups=B, dilation=dilation)
    return out.view(B, Cout, out.shape[2], out.shape[3])
