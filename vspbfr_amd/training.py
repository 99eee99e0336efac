"""Training-mode (differentiable) forward of `Restoration_net` over the gfx950 operators -- the generator half of the training step
(reference restoration_train.py:153-255 runs `generator(low_imgs, feats, latents, noise)` under autograd).

The inference modules (restorenet.py / layers.py) run fused, `no_grad` launches: style scaling inside the conv staging, demodulation,
noise, bias and activation in the conv epilogue -- none of which records a graph.  This module evaluates the SAME parameters
(same module tree, same state-dict keys) the way the reference's own forward does, op by op, with every op differentiable:

    convolutions      op.conv2d_gradfix.conv2d / conv_transpose2d   (data gradient on the forward kernels, weight gradient
                                                                     vsp_conv2d_wgrad_f32; the per-sample weights of a modulated
                                                                     layer go through the `groups = batch` form, as in
                                                                     models/RestoreNet.py:373-416)
    bias + leaky-ReLU op.fused_leaky_relu                            (vsp_fused_bias_act_f32, any-order autograd)
    blur / upsample   op.upfirdn2d                                   (vsp_upfirdn2d_f32, adjoint = upfirdn2d with flipped taps)
    everything else   torch tensor algebra on the device (weight modulation / demodulation, F.linear, concatenations) -- glue
                      whose gradients autograd derives; F.linear is a plain library GEMM.

`tests/test_hip_models.py::test_restorenet64_training_gradients` compares output and parameter / input gradients with the
reference's own backward pass (tests/golden/restorenet64_grad.npz, tools/make_golden.py::gen_restorenet64_grad)."""
import torch
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import hip_ops as H
from .op import conv2d_gradfix, fused_leaky_relu, upfirdn2d


_ONES = {}


class _EqualLinearGemm(Function):
    """scale * x @ W^T (+ lr_mul * bias) with the equalized-lr factors as the alpha / beta of the GEMM calls: one launch forward, three
    backward (the torch expression `F.linear(x, W * scale, bias * lr_mul)` is 3 + 5: ~50 such layers per iteration)."""

    @staticmethod
    def forward(ctx, x, weight, bias, scale, lr_mul):
        x2 = x.reshape(-1, x.shape[-1])
        if bias is not None:
            y = torch.addmm(bias, x2, weight.t(), beta=lr_mul, alpha=scale)
        else:
            y = torch.addmm(x2.new_empty((1, weight.shape[0])), x2, weight.t(), beta=0, alpha=scale)   # (beta = 0: the input is not read)
        ctx.save_for_backward(x2, weight)
        ctx.scale, ctx.lr_mul, ctx.xshape = scale, lr_mul, x.shape
        return y.view(*x.shape[:-1], weight.shape[0])

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x2, weight = ctx.saved_tensors
        g2 = g.reshape(-1, g.shape[-1]).contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.addmm(g2.new_empty((1, weight.shape[1])), g2, weight, beta=0, alpha=ctx.scale).view(ctx.xshape)
        if ctx.needs_input_grad[1]:
            dw = torch.addmm(g2.new_empty((1, weight.shape[1])), g2.t(), x2, beta=0, alpha=ctx.scale)
        if ctx.needs_input_grad[2]:
            key = (g2.device, g2.shape[0])
            if key not in _ONES:
                _ONES[key] = torch.ones(g2.shape[0], device=g2.device)
            db = torch.addmv(g2.new_empty(g2.shape[1]), g2.t(), _ONES[key], beta=0, alpha=ctx.lr_mul)
        return dx, dw, db, None, None


def equal_linear(x, lin, twice_differentiable=False):
    """EqualLinear.forward (reference models/RestoreNet.py:161-171).  `twice_differentiable`: the discriminator's layers (the R1 penalty
    differentiates through their gradient) keep the torch expression."""
    if H.FUSED_DEMOD_GRAD and x.is_cuda and not twice_differentiable:
        if lin.activation:
            return fused_leaky_relu(_EqualLinearGemm.apply(x, lin.weight, None, lin.scale, lin.lr_mul), lin.bias * lin.lr_mul)
        return _EqualLinearGemm.apply(x, lin.weight, lin.bias, lin.scale, lin.lr_mul)
    if lin.activation:
        return fused_leaky_relu(F.linear(x, lin.weight * lin.scale), lin.bias * lin.lr_mul)
    return F.linear(x, lin.weight * lin.scale, bias=None if lin.bias is None else lin.bias * lin.lr_mul)


def _blur(x, blur):
    return upfirdn2d(x, blur.kernel, pad=blur.pad)


def modulated_conv_grouped(x, conv, style, modulation=None):
    """ModulatedConv2d.forward, fused branch, LITERALLY as the reference evaluates it (models/RestoreNet.py:373-416): per-sample
    weights w[b] = scale * W * s[b], demodulated, applied as ONE grouped convolution with groups = batch.  Kept as the statement the
    fused form below is checked against (tests); the training step does not use it: it materialises B weight copies per layer and
    runs every layer at batch 1 per group."""
    B, cin, H, W = x.shape
    k, cout = conv.kernel_size, conv.out_channel
    s = equal_linear(style, modulation if modulation is not None else conv.modulation).view(B, 1, cin, 1, 1)
    w = conv.scale * conv.weight * s                                         # (B, Cout, Cin, k, k)
    if conv.demodulate:
        w = w * torch.rsqrt(w.pow(2).sum([2, 3, 4]) + 1e-8).view(B, cout, 1, 1, 1)
    if conv.upsample:
        wt = w.transpose(1, 2).reshape(B * cin, cout, k, k)
        out = conv2d_gradfix.conv_transpose2d(x.reshape(1, B * cin, H, W), wt, padding=0, stride=2, groups=B)
        return _blur(out.view(B, cout, out.shape[2], out.shape[3]), conv.blur)
    if conv.downsample:
        x = _blur(x, conv.blur)
        H, W = x.shape[2], x.shape[3]
        out = conv2d_gradfix.conv2d(x.reshape(1, B * cin, H, W), w.view(B * cout, cin, k, k), padding=0, stride=2, groups=B)
    else:
        out = conv2d_gradfix.conv2d(x.reshape(1, B * cin, H, W), w.view(B * cout, cin, k, k), padding=conv.padding,
                                    dilation=conv.dilation, groups=B)
    return out.view(B, cout, out.shape[2], out.shape[3])


# ---------------------------------------------------------------------------------------------------------------------------------
# Modulated convolution in the modulate-input / demodulate-output form:  y[b] = demod[b, co] * conv(x[b] * s[b, ci], scale * W).
# Identical to the reference's per-sample weights w[b] = scale W s[b] demod[b] (the two scales commute with the convolution), but ONE
# convolution over the whole batch with the SHARED weight: the tuned / Winograd kernels of the inference path with their fused
# in_scale / out_scale operands, a data gradient on the same kernels (scales exchanged), and ONE weight gradient with K = B * pixels
# (vsp_conv2d_wgrad_f32 folds both scales in) instead of B per-sample ones.  s and demod stay torch tensors, so the gradient reaches
# the style MLP, the modulation layer and -- through demod -- the weight by plain autograd; the Function below supplies d/dx,
# d/dW (direct term), d/ds (direct term) and d/d demod.  First order only: the generator is never differentiated twice
# (the R1 penalty runs through the discriminator, which has no modulated layers).
# ---------------------------------------------------------------------------------------------------------------------------------
def _adjoint_pack(layer):
    """Packed weight of the data gradient of `layer` (cached per weight version on the layer)."""
    def build():
        k, d = layer.kernel_size, layer.dilation
        if layer.upsample:       # forward = transposed conv; adjoint = stride-2 conv of g, weight read as (out = ci, in = co)
            return H.PackedConv(H.pack_weight(layer.weight[0], adjoint=True, scale=layer.scale), 1, layer.in_channel, layer.out_channel,
                                3, 3, 2, (1,), (0,))
        if layer.downsample:     # forward = stride-2 conv; adjoint = one-pass transposed conv (ordinary 3x3 packing of (out = ci, in = co))
            return H.PackedConv(H.pack_weight(layer.weight[0], adjoint=True, scale=layer.scale), 1, layer.in_channel, layer.out_channel,
                                3, 3, 1, (1,), (1,))
        # stride 1: correlation with the flipped kernel, channels exchanged
        return H.PackedConv(H.pack_weight(layer.weight[0], adjoint=True, flip=True, scale=layer.scale), 1, layer.in_channel,
                            layer.out_channel, k, k, 1, (d,), (d * (k - 1) - layer.padding,))
    return layer._derive("adjoint", [layer.weight], build)


class _ModConv(Function):
    """`inner_demod` (training default): the demodulation coefficients are computed HERE from (weight, s) and their gradient is
    ADDED to the convolution's own style / weight gradients inside the backward kernels (vsp_demod_weight_bwd_acc_f32), so the layer
    is one autograd node with one gradient per operand -- as a separate node (`_Demod`) the engine added two pairs of tensors per
    layer and copied a strided slice (148 adds and 60 copies per iteration)."""

    @staticmethod
    def forward(ctx, x, weight, s, demod, layer, inner_demod=False):
        x, s = x.contiguous(), s.contiguous()
        wsq = None
        if inner_demod:
            demod, wsq = H.demod_weight(s, weight, layer.scale)
        demod = None if demod is None else demod.contiguous()
        ctx.inner = bool(inner_demod)
        if layer.upsample:
            y = H.conv_transpose2d_s2_fused(x, layer.packed(), in_scale=s, out_scale=demod)
        else:
            y = H.conv2d_packed(x, layer.packed(), in_scale=s, out_scale=demod)
        ctx.layer = layer
        ctx.save_for_backward(x, s, demod, y if demod is not None else None, wsq, weight if inner_demod else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, s, demod, y, wsq, weight = ctx.saved_tensors
        layer = ctx.layer
        g = g.contiguous()
        B, cin, cout, k = x.shape[0], layer.in_channel, layer.out_channel, layer.kernel_size
        d_demod = None
        if demod is not None:
            d_demod = H.plane_dot(g, y) / demod                       # y = demod * raw  ->  d/d demod = <g, raw>
        dx = ds = dw = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[2]:
            adj = _adjoint_pack(layer)
            if layer.downsample:
                dxs = H.conv_transpose2d_s2_into(g, adj, x.shape[2:], in_scale=demod)   # (2 OH + 1)^2 (+ a zero row / column: H even)
            else:
                dxs = H.conv2d_packed(g, adj, in_scale=demod)
            ds = H.plane_dot_scale_(dxs, x, s)                        # dxs = d/d(x s): ds = <dxs, x>, then dxs *= s in the same pass
            dx = dxs
        if ctx.needs_input_grad[1] and not conv2d_gradfix.weight_gradients_disabled:
            if layer.upsample:   # the stride-2 weight gradient with the roles of x and g exchanged; result in (Cin, Cout, 3, 3)
                dw = H.conv2d_wgrad(g, x, (cin, cout, 3, 3), 2, 0, 1, 1, x_scale=demod, dy_scale=s, scale=layer.scale).transpose(0, 1)
            else:
                dw = H.conv2d_wgrad(x, g, (cout, cin, k, k), 2 if layer.downsample else 1, 0 if layer.downsample else layer.padding,
                                    layer.dilation, 1, x_scale=s, dy_scale=demod, scale=layer.scale)
            dw = dw.unsqueeze(0)   # (the equalized-lr factor is folded into the kernel's store)
        if ctx.inner:   # the demodulation's share of d/ds and d/dW, added in place
            need_s, need_w = ctx.needs_input_grad[2], ctx.needs_input_grad[1]
            if dw is not None and not dw.is_contiguous():
                dw = dw.contiguous()                                   # (up-sampling layers: the exchanged-roles gradient is a transposed view)
            if (not need_s or ds is not None) and (not need_w or dw is not None):
                H.demod_weight_bwd(d_demod, demod, s, wsq, weight, layer.scale, need_s, need_w, ds_out=ds if need_s else None,
                                   dw_out=dw if need_w else None, accumulate=True)
            else:   # (a gradient the convolution did not produce: weight gradients disabled)
                ds2, dw2 = H.demod_weight_bwd(d_demod, demod, s, wsq, weight, layer.scale, need_s, need_w)
                if need_s:
                    ds = ds2 if ds is None else ds + ds2
                if need_w:
                    dw = dw2 if dw is None else dw + dw2
            return dx, dw, ds, None, None, None
        return dx, dw, ds, d_demod, None, None


class _Demod(Function):
    """rsqrt(scale^2 * s^2 @ (sum_k W^2)^T + 1e-8) and its first-order gradient in s and W as three launches
    (vsp_demod_weight_f32 / _bwd_f32) instead of the ~20 small ones of the torch expression below."""

    @staticmethod
    def forward(ctx, weight, s, scale):
        s = s.contiguous()
        out, wsq = H.demod_weight(s, weight, scale)
        ctx.save_for_backward(weight, s, wsq, out)
        ctx.scale = scale
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        weight, s, wsq, out = ctx.saved_tensors
        ds, dw = H.demod_weight_bwd(g.contiguous(), out, s, wsq, weight, ctx.scale, ctx.needs_input_grad[1], ctx.needs_input_grad[0])
        return dw, ds, None


def _demod(conv, s):
    """rsqrt(sum_{ci,k} (scale W s)^2 + 1e-8) = rsqrt(s^2 @ (scale^2 sum_k W^2)^T + 1e-8): (B, Cout), differentiable in s and W."""
    if H.FUSED_DEMOD_GRAD and s.shape[0] <= 16 and s.is_cuda:
        return _Demod.apply(conv.weight, s, conv.scale)
    wsq = conv.weight[0].pow(2).sum((2, 3))
    return torch.rsqrt(F.linear(s * s, wsq) * (conv.scale ** 2) + 1e-8)


def modulated_conv(x, conv, style, modulation=None):
    """ModulatedConv2d.forward (reference models/RestoreNet.py:373-416) in the fused form above."""
    s = equal_linear(style, modulation if modulation is not None else conv.modulation)
    inner = bool(conv.demodulate and H.FUSED_DEMOD_GRAD and s.is_cuda and s.shape[0] <= 16)
    demod = _demod(conv, s) if (conv.demodulate and not inner) else None
    if conv.downsample:
        x = _blur(x, conv.blur)
    if torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad or s.requires_grad):
        out = _ModConv.apply(x, conv.weight, s, demod, conv, inner)
    else:
        out = _ModConv.forward(_NoCtx(), x, conv.weight, s, demod, conv, inner)
    return _blur(out, conv.blur) if conv.upsample else out


class _NoCtx:
    inner = False

    def save_for_backward(self, *a):
        pass


# ---- the four dilated branches of a SMART layer as ONE launch (shared input and modulation, per-branch weight / demodulation)
def _smart_adjoint_pack(layer, hw=None):
    """Packed weight of the data gradient of the four branches.  Maps of 64 x 64 and larger (`hw`): ONE convolution over all nb * cg
    gradient channels whose dilation follows the input-channel quarter (vsp_conv_params.dil_by_input_quarter) -- no (B, nb * Cin, H, W)
    intermediate, no sum over the branches; smaller maps (the 256-pixel tiles of the pipelined kernel leave the chip under-filled: 512 channels at 32^2 326 vs 175 us): a true grouped
    conv, group i = branch i, summed by the caller."""
    ws = [m.weight for m in layer.ModulatedConv2ds]
    m0 = layer.ModulatedConv2ds[0]
    cg, cin = layer.out_channel // len(ws), layer.in_channel
    dil = tuple(m.dilation for m in layer.ModulatedConv2ds)
    pad = tuple(m.dilation * (m.kernel_size - 1) - m.padding for m in layer.ModulatedConv2ds)

    def build():
        # true grouped conv over g: group i reads its branch's cg channels, writes Cin channels, dilation / padding of branch i
        wp = H.pack_weight_stack([w[0] for w in ws], adjoint=True, flip=True, scale=m0.scale)
        return H.PackedConv(wp, len(ws), layer.in_channel, cg, 3, 3, 1, dil, pad, x_group_stride=cg)

    def build_q():
        wp = H.pack_weight_stack([w[0] for w in ws], adjoint=True, flip=True, scale=m0.scale)   # [branch][tap][cg][Cin]
        # weight image: four blocks of Cin / 4 output channels, [block][tap][q * cg + c][Cin / 4]
        a = wp.permute(1, 0, 2, 3).reshape(9, len(ws) * cg, cin)
        w2 = a.view(9, len(ws) * cg, 4, cin // 4).permute(2, 0, 1, 3).contiguous()
        return H.PackedConv(w2, 4, cin // 4, len(ws) * cg, 3, 3, 1, dil, pad, dil_by_input_quarter=True)

    one_pass = (H.SMART_ADJOINT_ONE_PASS and hw is not None and min(hw) >= 64 and len(ws) == 4 and cg % 4 == 0 and cin % 16 == 0
                and pad == dil)
    return layer._derive("branches_adjoint_q", ws, build_q) if one_pass else layer._derive("branches_adjoint", ws, build)


class _SmartBranches(Function):
    """demod = None: the four demodulation vectors are computed here and their gradients are added inside the backward kernels (see _ModConv)."""

    @staticmethod
    def forward(ctx, x, s, demod, layer, *weights):
        x, s = x.contiguous(), s.contiguous()
        ctx.inner = demod is None
        wsqs = []
        if demod is None:
            parts = []
            for m in layer.ModulatedConv2ds:
                dm, wq = H.demod_weight(s, m.weight, m.scale)
                parts.append(dm)
                wsqs.append(wq)
            demod = torch.cat(parts, 1)
        demod = demod.contiguous()
        ctx.n_w = len(weights)
        pc = layer._branch_pack()
        y = H.conv2d_packed(x, pc, in_scale=s, out_scale=demod)
        ctx.layer = layer
        ctx.save_for_backward(x, s, demod, y, *wsqs, *(weights if wsqs else ()))
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, s, demod, y = ctx.saved_tensors[:4]
        wsqs = ctx.saved_tensors[4:4 + ctx.n_w] if ctx.inner else ()
        wts = ctx.saved_tensors[4 + ctx.n_w:] if ctx.inner else ()
        layer = ctx.layer
        g = g.contiguous()
        B, cin, Hh, Ww = x.shape
        nb = len(layer.ModulatedConv2ds)
        cg = layer.out_channel // nb
        d_demod = H.plane_dot(g, y) / demod
        dx = ds = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            adj = _smart_adjoint_pack(layer, (Hh, Ww))
            if adj.dil_by_input_quarter:
                dxs = H.conv2d_packed(g, adj, in_scale=demod)                                # (B, Cin, H, W): the branches summed in the K loop
            else:
                parts = H.conv2d_packed(g, adj, in_scale=demod)                              # (B, nb * Cin, H, W)
                dxs = parts.view(B, nb, cin, Hh, Ww).sum(1)
            ds = H.plane_dot_scale_(dxs, x, s)
            dx = dxs
        dws = [None] * nb
        if any(ctx.needs_input_grad[4:]) and not conv2d_gradfix.weight_gradients_disabled:
            ms = layer.ModulatedConv2ds   # one launch: 4 groups over the shared input, each with its branch's dilation
            dw = H.conv2d_wgrad(x, g, (nb * cg, cin, 3, 3), 1, tuple(m.padding for m in ms), tuple(m.dilation for m in ms), nb,
                                x_scale=s, dy_scale=demod, x_shared=True, scale=ms[0].scale)   # (equal shapes: one factor)
            for i, m in enumerate(ms):
                dws[i] = dw[i * cg:(i + 1) * cg].unsqueeze(0)
        if ctx.inner:
            for i, m in enumerate(layer.ModulatedConv2ds):
                need_s, need_w = ctx.needs_input_grad[1], ctx.needs_input_grad[4 + i]
                gi, oi = d_demod[:, i * cg:(i + 1) * cg], demod[:, i * cg:(i + 1) * cg]
                if (ds is not None or not need_s) and (dws[i] is not None or not need_w):
                    H.demod_weight_bwd(gi, oi, s, wsqs[i], wts[i], m.scale, need_s, need_w, ds_out=ds if need_s else None,
                                       dw_out=dws[i] if need_w else None, accumulate=True)
                else:   # (a gradient the convolution did not produce: weight gradients disabled)
                    ds2, dw2 = H.demod_weight_bwd(gi, oi, s, wsqs[i], wts[i], m.scale, need_s, need_w)
                    ds = ds2 if (need_s and ds is None) else (ds + ds2 if need_s else ds)
                    dws[i] = dw2 if (need_w and dws[i] is None) else (dws[i] + dw2 if need_w else dws[i])
            return (dx, ds, None, None, *dws)
        return (dx, ds, d_demod, None, *dws)


def smart_branches(x, layer, style):
    s = equal_linear(style, layer.modulation)
    inner = bool(H.FUSED_DEMOD_GRAD and s.is_cuda and s.shape[0] <= 16)
    demod = None if inner else torch.cat([_demod(m, s) for m in layer.ModulatedConv2ds], 1)
    ws = [m.weight for m in layer.ModulatedConv2ds]
    if torch.is_grad_enabled() and (x.requires_grad or s.requires_grad or any(w.requires_grad for w in ws)):
        return _SmartBranches.apply(x, s, demod, layer, *ws)
    return _SmartBranches.forward(_NoCtx(), x, s, demod, layer, *ws)


class _NoiseBiasAct(Function):
    """NoiseInjection + FusedLeakyReLU as one launch; backward = the slope mask of fused_bias_act (from y), the channel sum for
    the bias and one dot product for the scalar noise weight.  First order (generator only)."""

    @staticmethod
    def forward(ctx, x, noise, noise_w, bias):
        noise = noise.contiguous()
        y = H.noise_bias_act(x.contiguous(), noise, noise_w.contiguous(), bias.contiguous())
        ctx.save_for_backward(y, noise)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        y, noise = ctx.saved_tensors
        gx = H.fused_bias_act(g.contiguous(), g.new_empty(0), y, 3, 1, 0.2, 2 ** 0.5)
        dnw = H.noise_dot(gx, noise) if ctx.needs_input_grad[2] else None
        db = H.channel_sum(gx) if ctx.needs_input_grad[3] else None
        return gx, None, dnw, db


def noise_bias_act(x, noise, noise_w, bias):
    """out = x + noise_w * noise; fused_leaky_relu(out, bias)  (NoiseInjection.forward + FusedLeakyReLU.forward)."""
    if torch.is_grad_enabled() and (x.requires_grad or noise_w.requires_grad or bias.requires_grad):
        return _NoiseBiasAct.apply(x, noise, noise_w, bias)
    return H.noise_bias_act(x.contiguous(), noise.contiguous(), noise_w.contiguous(), bias.contiguous())


def styled_conv(x, layer, style, noise):
    """StyledConv.forward (reference models/RestoreNet.py:599-603): conv -> noise -> FusedLeakyReLU."""
    out = modulated_conv(x, layer.conv, style)
    return noise_bias_act(out, noise, layer.noise.weight, layer.activate.bias)


class _FusionTail(Function):
    """The fusion conv of a SMART layer with its tail -- FusedLeakyReLU, NoiseInjection, FusedLeakyReLU -- in the conv's epilogue (the
    launch the inference module makes), and ONE backward stream for the tail (vsp_smart_tail_bwd_f32: both masks from the final y, the
    two bias gradients and the noise-weight gradient) in front of the data- and weight-gradient convs.  First order (generator only)."""

    @staticmethod
    def forward(ctx, x, weight, bias1, noise, noise_w, bias2, layer):
        x, noise = x.contiguous(), noise.contiguous()
        y = H.conv2d_packed(x, layer._fusion_pack(), act1=True, bias1=bias1, noise=noise, noise_w=noise_w, act2=1, bias2=bias2)
        ctx.layer = layer
        ctx.save_for_backward(x, y, noise, noise_w, bias2)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, y, noise, noise_w, bias2 = ctx.saved_tensors
        layer = ctx.layer
        conv = layer.fusion[0]
        g1, db1, db2, dnw = H.smart_tail_bwd(g.contiguous(), y, noise, noise_w, bias2)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            c = layer.out_channel
            adj = layer._derive("fusion_adjoint", [conv.weight], lambda: H.PackedConv(
                H.pack_weight(conv.weight, adjoint=True, flip=True, scale=conv.scale), 1, c, c, 3, 3, 1, (1,), (1,)))
            dx = H.conv2d_packed(g1, adj)
        if ctx.needs_input_grad[1] and not conv2d_gradfix.weight_gradients_disabled:
            dw = H.conv2d_wgrad(x, g1, tuple(conv.weight.shape), 1, 1, 1, 1, scale=conv.scale)
        return dx, dw, db1, None, dnw, db2, None


def smart_layer(x, layer, style, noise):
    """SMART_layer.forward (reference models/RestoreNet.py:220-244): the four dilated branches share ONE modulation; fusion conv,
    FusedLeakyReLU, noise, FusedLeakyReLU."""
    out = smart_branches(x, layer, style)
    f = layer.fusion[0]
    if torch.is_grad_enabled():
        return _FusionTail.apply(out, f.weight, layer.fusion[1].bias, noise, layer.noise.weight, layer.activate.bias, layer)
    out = conv2d_gradfix.conv2d(out, f.weight * f.scale, padding=1)
    out = fused_leaky_relu(out, layer.fusion[1].bias)
    return noise_bias_act(out, noise, layer.noise.weight, layer.activate.bias)


def large_conv_layer(x, layer):
    """LargeConvLayer.forward (reference models/RestoreNet.py:773-787)."""
    outs = [conv2d_gradfix.conv2d(x, m.weight * m.scale, padding=m.padding, dilation=m.dilation) for m in layer.dilated_convs]
    f = layer.fusion[0]
    out = conv2d_gradfix.conv2d(torch.cat(outs, dim=1), f.weight * f.scale, padding=f.padding)
    out = fused_leaky_relu(out, layer.fusion[1].bias)
    return fused_leaky_relu(out, layer.activate.bias)


def to_rgb(x, layer, style, skip=None):
    """ToRGB.forward (reference models/RestoreNet.py:657-666)."""
    out = modulated_conv(x, layer.conv, style) + layer.bias
    if skip is not None:
        u = layer.upsample
        out = out + upfirdn2d(skip, u.kernel, up=u.factor, down=1, pad=u.pad)
    return out


class _Add3(Function):
    """a + b + c in one stream (the decoder's `out + features[k] + de_feats[k]`, models/RestoreNet.py:1035); the gradient is the
    incoming one for every operand that asks."""

    @staticmethod
    def forward(ctx, a, b, c):
        return H.add3(a.contiguous(), b.contiguous(), c.contiguous())

    @staticmethod
    def backward(ctx, g):
        return tuple(g if need else None for need in ctx.needs_input_grad)


def restoration_net_forward(net, images, de_feats, pre_styles, noise_styles, enc_noise, dec_noise, inject_index=None):
    """Restoration_net.forward + encoder_forward (reference models/RestoreNet.py:915-942, 968-1046) with explicit noise maps
    (enc_noise: 2 per encoder level, dec_noise: 1 + 2 per decoder level, as the inference module takes them).  Dropout2d of
    `final_linear` follows the module's mode (identity in eval)."""
    B = images.shape[0]
    styles = []
    for z in noise_styles:
        h = z * torch.rsqrt(torch.mean(z ** 2, dim=1, keepdim=True) + 1e-8)                 # PixelNorm
        for lin in list(net.style)[1:]:
            h = equal_linear(h, lin)
        styles.append(h)
    if len(styles) < 2:
        noise_latent = styles[0].unsqueeze(1).repeat(1, net.n_latent, 1)
    else:
        if inject_index is None:
            raise RuntimeError("two noise codes need an explicit inject_index")
        noise_latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                                  styles[1].unsqueeze(1).repeat(1, net.n_latent - inject_index, 1)], 1)
    latent = torch.cat([pre_styles[:, :net.n_latent], noise_latent], dim=-1)
    latent_cp = torch.flip(latent, dims=[1])

    out = large_conv_layer(images, net.down_from_big)
    feats = []
    for ii in range(0, len(net.encoder_convs), 2):
        out = smart_layer(out, net.encoder_convs[ii], latent_cp[:, ii], enc_noise[ii])
        feats.append(out)
        out = styled_conv(out, net.encoder_convs[ii + 1], latent_cp[:, ii], enc_noise[ii + 1])
    out = large_conv_layer(out, net.final_layer)
    x_global = net.final_linear[1](equal_linear(out.reshape(B, -1), net.final_linear[0]))
    early = equal_linear(x_global, net.final_transfer)
    feats.append(out + early.view(B, -1, 4, 4))
    feats = feats[::-1]

    def sty(i):
        return torch.cat([latent[:, i], x_global], dim=1)

    out = smart_layer(feats[0], net.conv1, sty(0), dec_noise[0])
    skip = to_rgb(out, net.to_rgb1, sty(1))
    i = 1
    for j in range(net.log_size - 2):
        out = styled_conv(out, net.convs[2 * j], sty(i), dec_noise[1 + 2 * j])
        k = (i + 1) // 2
        out = _Add3.apply(out, feats[k], de_feats[k])
        out = smart_layer(out, net.convs[2 * j + 1], sty(i + 1), dec_noise[2 + 2 * j])
        skip = to_rgb(out, net.to_rgbs[j], sty(i + 2), skip)
        i += 2
    return skip



# ---------------------------------------------------------------------------------------------------------------------------------
# Stage-B training (reference code_diffuser_train.py:153-190): the Code_diffuser is trained THROUGH the frozen StyleGAN2 prior.
#   * the denoiser (four TACC blocks, 18 x 512 tokens, ~0.3 GFLOP per block) runs as torch tensor algebra -- F.linear / matmul are
#     plain library GEMMs, the rest is a handful of row operations -- because every one of its 72 tensors needs a gradient and the
#     arithmetic is negligible beside the decoder's; the fused chain of the inference path (tacc_chain.hip) records no graph;
#   * the prior's synthesis network runs on the modulated-convolution Functions above (data gradient and style gradients only: the
#     decoder is frozen -- the reference switches its requires_grad on, code_diffuser_train.py:166, but never steps it).
# ---------------------------------------------------------------------------------------------------------------------------------
def _scaled_lrelu(x):
    return F.leaky_relu(x, 0.2) * (2 ** 0.5)


def _head(head, c):
    """gamma_ / beta_ of a TACC block (models/CodeDiffuser.py:77-78): Linear(513, 512), LayerNorm, ScaledLeakyReLU, Linear, Sigmoid | ScaledLeakyReLU."""
    h = F.linear(c, head[0].weight, head[0].bias)
    h = _scaled_lrelu(F.layer_norm(h, (h.shape[-1],), head[1].weight, head[1].bias, 1e-5))
    h = F.linear(h, head[3].weight, head[3].bias)
    return torch.sigmoid(h) if head.last == 2 else _scaled_lrelu(h)


def tacc_block_forward(blk, x, embd, step):
    """TACC_block.forward (models/CodeDiffuser.py:86-116) with spatial_attention.forward (:30-47), differentiable."""
    d = x.shape[-1]
    x = x * torch.rsqrt(torch.mean(x ** 2, dim=1, keepdim=True) + 1e-8)           # PixelNorm over the 18 tokens
    K, V = F.linear(x, blk.k_matrix.weight), F.linear(x, blk.v_matrix.weight)
    c = torch.cat([embd, step], dim=-1)
    Q = F.linear(c, blk.q_matrix.weight).permute(0, 2, 1)
    h = torch.matmul(F.softmax(torch.matmul(K, Q) / (blk.dk ** 0.5), dim=-1), V)
    a = blk.attention_layer
    q, v = F.linear(x, a.q_matrix.weight), F.linear(x, a.v_matrix.weight)
    # spatial_attention: out = v @ softmax(k^T q / sqrt(d), dim=1).  Evaluated on the TRANSPOSED logits (q^T k, softmax over the LAST
    # dim, v @ A'^T): the same numbers, but torch's softmax over an inner dim of a (B, 512, 512) tensor is 10x slower than over the last
    kc = F.linear(c, a.k_matrix.weight)                                                  # (B, 18, 512)
    att_t = F.softmax(torch.matmul(q.transpose(1, 2), kc) / (d ** 0.5), dim=-1)          # (B, c', c) = softmax_c(score[c, c'])
    t = torch.matmul(v, att_t.transpose(1, 2))
    t = F.layer_norm(t, (d,), None, None, 1e-5)
    h = F.layer_norm(h + t, (d,), None, None, 1e-5)
    return h * (1.0 + _head(blk.gamma_, c)) + _head(blk.beta_, c)


def code_diffuser_forward(net, x, embd, t):
    """Code_diffuser.forward (models/CodeDiffuser.py:133-140)."""
    step = (t.float() / net.max_period).view(-1, 1, 1).repeat(1, embd.shape[1], 1)
    for blk in net.att_mapper:
        x = tacc_block_forward(blk, x, embd, step)
    return x


def ddpm_training_forward(ddpm, x, condi_in, noise=None):
    """My_DDPM.forward(training=True) (ldm/ddpm.py:412-421): q_sample at t = T - 1, then the T posterior-mean steps; returns the
    last denoised codes and the list [x_noisy, x_{T-1}, ..., x_0].  `noise`: the q_sample draw (default torch.randn_like)."""
    if ddpm.parameterization != "x0" or ddpm.clip_denoised:
        raise RuntimeError("ddpm_training_forward: the x0-parameterised, unclipped sampler of code_diffuser_train.py")
    B, T = condi_in.shape[0], ddpm.num_timesteps
    noise = torch.randn_like(x) if noise is None else noise
    last = ddpm.sqrt_alphas_cumprod[T - 1] * x + ddpm.sqrt_one_minus_alphas_cumprod[T - 1] * noise
    seq = [last]
    for i in reversed(range(T)):
        t = torch.full((B,), i, device=x.device, dtype=torch.long)
        x0 = code_diffuser_forward(ddpm.model, last, condi_in, t)
        last = ddpm.posterior_mean_coef1[i] * x0 + ddpm.posterior_mean_coef2[i] * last
        seq.append(last)
    return last, seq


class KDLoss(torch.nn.Module):
    """code_diffuser_train.py:64-90: (KL(softmax(S1 / T) || softmax(S2 / T)) batch mean, L1(S2, S1)) summed over the list, S1 detached."""

    def __init__(self, loss_weight=1.0, temperature=0.15):
        super().__init__()
        self.loss_weight, self.temperature = loss_weight, temperature

    def forward(self, S1_fea, S2_fea):
        dis = ab = 0
        for s1, s2 in zip(S1_fea, S2_fea):
            s1 = s1.detach()
            dis = dis + F.kl_div(F.log_softmax(s2 / self.temperature, dim=1), F.softmax(s1 / self.temperature, dim=1), reduction="batchmean")
            ab = ab + (s2 - s1).abs().mean()
        return self.loss_weight * dis, self.loss_weight * ab


def generator_forward(gen, latent, noise=None):
    """e4e Generator.forward([latent], input_is_latent=True) (e4e/models/stylegan2/model.py:475-552) -> image, differentiable in
    `latent` (B, n_latent, 512) (and in the decoder's parameters where they require a gradient).  noise: one map per layer
    (default: fresh normal draws, randomize_noise=True)."""
    B = latent.shape[0]
    if noise is None:
        noise = [torch.randn(B, 1, 2 ** ((i + 5) // 2), 2 ** ((i + 5) // 2), device=latent.device) for i in range(gen.num_layers)]
    out = gen.input.input.repeat(B, 1, 1, 1)
    out = styled_conv(out, gen.conv1, latent[:, 0], noise[0])
    skip = to_rgb(out, gen.to_rgb1, latent[:, 1])
    i = 1
    for j in range(gen.log_size - 2):
        out = styled_conv(out, gen.convs[2 * j], latent[:, i], noise[1 + 2 * j])
        out = styled_conv(out, gen.convs[2 * j + 1], latent[:, i + 1], noise[2 + 2 * j])
        skip = to_rgb(out, gen.to_rgbs[j], latent[:, i + 2], skip)
        i += 2
    return skip


class _AvgPool2(Function):
    @staticmethod
    def forward(ctx, x):
        return H.avgpool2x2(x.contiguous())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return g.repeat_interleave(2, 2).repeat_interleave(2, 3).mul_(0.25)


def face_pool(img, out_size):
    """AdaptiveAvgPool2d((out_size, out_size)) of e4e/models/psp.py:101 for the power-of-two ratios the path uses (exact 2x2 means)."""
    while img.shape[-1] > out_size:
        if img.shape[-1] % 2 or img.shape[-1] // 2 < out_size:
            raise RuntimeError("face_pool: output size must divide the image size by a power of two")
        img = _AvgPool2.apply(img) if (torch.is_grad_enabled() and img.requires_grad) else H.avgpool2x2(img.contiguous())
    return img
