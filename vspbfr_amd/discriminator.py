"""`Discriminator` of the restoration GAN (reference models/RestoreNet.py:1137-1265) and the adversarial losses of the training
step (restoration_train.py:60-79) over the gfx950 operators -- SURVEY 8f rows 2 and 4.

Same constructor signature, same state-dict keys / shapes as the reference (`ConvLayer` = Sequential[Blur?, EqualConv2d,
FusedLeakyReLU?], `ResBlock` = conv1 / conv2 / skip: pinned by tests/test_layout.py against the reference's own module).  The
forward is differentiable to second order -- the R1 penalty differentiates the input gradient again -- because every operator on
the way is: op.conv2d_gradfix (forward / data-gradient kernels + vsp_conv2d_wgrad_f32), op.fused_leaky_relu, op.upfirdn2d, and
torch tensor algebra for the minibatch-stddev statistic and the two linear layers (plain library GEMMs).
ADA augmentation lives in vspbfr_amd/non_leaking.py."""
import contextlib
import math

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import hip_ops
from .layers import Blur, EqualConv2d, EqualLinear, LeakyBias
from .op import conv2d_gradfix, fused_leaky_relu, upfirdn2d
from .training import equal_linear


# The R1 penalty differentiates the discriminator's input gradient again, so by default every operator of the forward is twice
# differentiable.  The logistic-loss passes of the training step (restoration_train.py:196-212, 225-234) only need first order: inside
# `first_order()` an activated ConvLayer runs as ONE launch (bias + leaky ReLU in the conv epilogue, equalised-lr scale folded into
# the packing) with a hand-written backward -- slope mask from y, bias sum, data- and weight-gradient kernels.
_FIRST_ORDER = False


@contextlib.contextmanager
def first_order():
    global _FIRST_ORDER
    old, _FIRST_ORDER = _FIRST_ORDER, True
    try:
        yield
    finally:
        _FIRST_ORDER = old


def _cached_pack(conv, name, build):
    key = (conv.weight._version, conv.weight.data_ptr())
    cache = conv.__dict__.setdefault("_vsp_packs", {})
    hit = cache.get(name)
    if hit is None or hit[0] != key:
        hit = cache[name] = (key, build())
    return hit[1]


class _ConvLrelu(Function):
    """`gain`: the activation's gain (sqrt 2 of FusedLeakyReLU; 1 when the ResBlock's 1 / sqrt 2 is folded in)."""

    @staticmethod
    def forward(ctx, x, weight, bias, conv, gain=2 ** 0.5):
        x = x.contiguous()
        cout, cin, k, _ = weight.shape
        pc = _cached_pack(conv, "fwd", lambda: hip_ops.PackedConv(hip_ops.pack_weight(weight, scale=conv.scale), 1, cout, cin, k, k,
                                                                  conv.stride, (1,), (conv.padding,)))
        y = hip_ops.conv2d_packed(x, pc, act2=1, bias2=bias, gain2=gain)
        ctx.conv, ctx.gain = conv, gain
        ctx.save_for_backward(x, y, weight)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, y, weight = ctx.saved_tensors
        conv = ctx.conv
        cout, cin, k, _ = weight.shape
        g1 = hip_ops.fused_bias_act(g.contiguous(), g.new_empty(0), y, 3, 1, 0.2, ctx.gain)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if conv.stride == 1:
                adj = _cached_pack(conv, "adj", lambda: hip_ops.PackedConv(
                    hip_ops.pack_weight(weight, adjoint=True, flip=True, scale=conv.scale), 1, cin, cout, k, k, 1, (1,), (k - 1 - conv.padding,)))
                dx = hip_ops.conv2d_packed(g1, adj)
            else:   # 3x3, stride 2, padding 0 (behind the blur): the one-pass transposed kernel writes straight into the input-sized gradient
                adj = _cached_pack(conv, "adj", lambda: hip_ops.PackedConv(
                    hip_ops.pack_weight(weight, adjoint=True, scale=conv.scale), 1, cin, cout, 3, 3, 1, (1,), (1,)))
                dx = hip_ops.conv_transpose2d_s2_into(g1, adj, x.shape[2:])
        if ctx.needs_input_grad[1] and not conv2d_gradfix.weight_gradients_disabled:
            dw = hip_ops.conv2d_wgrad(x, g1, tuple(weight.shape), conv.stride, conv.padding, 1, 1, scale=conv.scale)
        if ctx.needs_input_grad[2]:
            db = hip_ops.channel_sum(g1)
        return dx, dw, db, None, None


class _SkipConvAdd(Function):
    """Tail of a ResBlock, (main + skip(x)) / sqrt 2 (models/RestoreNet.py:1196-1202), as the epilogue of the skip's 1x1 stride-2 conv:
    y = main + conv(xb, W scale / sqrt 2) with `main` already carrying its 1 / sqrt 2 (activation gain 1 instead of sqrt 2)."""

    @staticmethod
    def forward(ctx, xb, weight, main, conv):
        xb, main = xb.contiguous(), main.contiguous()
        cout, cin = weight.shape[:2]
        sc = conv.scale / math.sqrt(2)
        pc = _cached_pack(conv, "fwd", lambda: hip_ops.PackedConv(hip_ops.pack_weight(weight, scale=sc), 1, cout, cin, 1, 1, 2, (1,), (0,)))
        y = hip_ops.conv2d_packed(xb, pc, res1=main)
        ctx.conv = conv
        ctx.save_for_backward(xb, weight)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        xb, weight = ctx.saved_tensors
        conv = ctx.conv
        cout, cin = weight.shape[:2]
        sc = conv.scale / math.sqrt(2)
        g = g.contiguous()
        dxb = dw = None
        if ctx.needs_input_grad[0]:   # the 1x1 adjoint lands on the even pixels of the blurred input
            adj = _cached_pack(conv, "adj", lambda: hip_ops.PackedConv(hip_ops.pack_weight(weight, adjoint=True, scale=sc), 1, cin, cout,
                                                                       1, 1, 1, (1,), (0,)))
            dxb = torch.zeros_like(xb)
            hip_ops.conv2d_packed(g, adj, out=dxb, out_stride=(2, 2))
        if ctx.needs_input_grad[1] and not conv2d_gradfix.weight_gradients_disabled:
            dw = hip_ops.conv2d_wgrad(xb, g, tuple(weight.shape), 2, 0, 1, 1, scale=sc)
        return dxb, dw, g if ctx.needs_input_grad[2] else None, None


def _blur_run(t, blur):
    from .op.upfirdn2d import _run
    p = blur.pad
    return _run(t, blur.kernel, (1, 1), (1, 1), (p[0], p[1], p[0], p[1]))


def _blur_adjoint(g, blur, in_hw):
    """Gradient of _blur_run with respect to its input of spatial size in_hw (the flipped kernel is cached on the module)."""
    from .op.upfirdn2d import _run
    k = blur.kernel
    fk = blur.__dict__.get("_vsp_flipped")
    if fk is None or fk[0] != (k.data_ptr(), k._version):
        fk = blur.__dict__["_vsp_flipped"] = ((k.data_ptr(), k._version), torch.flip(k, [0, 1]).contiguous())
    kh, kw = k.shape
    p0 = blur.pad[0]
    oh, ow = g.shape[2:]
    return _run(g, fk[1], (1, 1), (1, 1), (kw - p0 - 1, in_hw[1] - ow + p0, kh - p0 - 1, in_hw[0] - oh + p0))


class _ResBlockFO(Function):
    """A whole ResBlock of the first-order passes as ONE autograd node (models/RestoreNet.py:1183-1202): conv1 + lrelu, blur, stride-2
    conv2 + lrelu (gain 1: the 1 / sqrt 2 folded in), blur + 1x1 stride-2 skip with the sum in its epilogue.  What the single node buys
    is the BACKWARD's last step: the gradient of the skip path is handed to conv1's data-gradient convolution as a residual operand,
    so the two gradients of the block input meet in that kernel's epilogue instead of an activation-sized autograd add (the largest
    ATen kernels of the iteration: 268 MB operands at the first block)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, ws, blk):
        x = x.contiguous()
        c1, c2, cs = blk.conv1[0], blk.conv2[1], blk.skip[1]
        blur2, blurs = blk.conv2[0], blk.skip[0]
        ci, co = w1.shape[1], w2.shape[0]
        p1 = _cached_pack(c1, "fwd", lambda: hip_ops.PackedConv(hip_ops.pack_weight(w1, scale=c1.scale), 1, ci, ci, 3, 3, 1, (1,), (c1.padding,)))
        y1 = hip_ops.conv2d_packed(x, p1, act2=1, bias2=b1, gain2=2 ** 0.5)
        y1b = _blur_run(y1, blur2)
        p2 = _cached_pack(c2, "fwd", lambda: hip_ops.PackedConv(hip_ops.pack_weight(w2, scale=c2.scale), 1, co, ci, 3, 3, 2, (1,), (0,)))
        main = hip_ops.conv2d_packed(y1b, p2, act2=1, bias2=b2, gain2=1.0)
        xb = _blur_run(x, blurs)
        sc = cs.scale / math.sqrt(2)
        ps = _cached_pack(cs, "fwd", lambda: hip_ops.PackedConv(hip_ops.pack_weight(ws, scale=sc), 1, co, ci, 1, 1, 2, (1,), (0,)))
        y = hip_ops.conv2d_packed(xb, ps, res1=main)
        ctx.blk = blk
        ctx.save_for_backward(x, y1, y1b, main, xb, w1, w2, ws)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, y1, y1b, main, xb, w1, w2, ws = ctx.saved_tensors
        blk = ctx.blk
        c1, c2, cs = blk.conv1[0], blk.conv2[1], blk.skip[1]
        blur2, blurs = blk.conv2[0], blk.skip[0]
        ci, co = w1.shape[1], w2.shape[0]
        sc = cs.scale / math.sqrt(2)
        g = g.contiguous()
        no_w = conv2d_gradfix.weight_gradients_disabled
        need = ctx.needs_input_grad
        dx = dw1 = db1 = dw2 = db2 = dws = None
        # ---- skip path: the 1x1 adjoint lands on the even pixels of the blurred input, then the blur's adjoint
        dx_skip = None
        if need[0]:
            adjs = _cached_pack(cs, "adj", lambda: hip_ops.PackedConv(hip_ops.pack_weight(ws, adjoint=True, scale=sc), 1, ci, co, 1, 1, 1, (1,), (0,)))
            dxb = torch.zeros_like(xb)
            hip_ops.conv2d_packed(g, adjs, out=dxb, out_stride=(2, 2))
            dx_skip = _blur_adjoint(dxb, blurs, x.shape[2:])
        if need[5] and not no_w:
            dws = hip_ops.conv2d_wgrad(xb, g, tuple(ws.shape), 2, 0, 1, 1, scale=sc)
        # ---- main path, last layer first
        g2 = hip_ops.fused_bias_act(g, g.new_empty(0), main, 3, 1, 0.2, 1.0)
        if need[3] and not no_w:
            dw2 = hip_ops.conv2d_wgrad(y1b, g2, tuple(w2.shape), 2, 0, 1, 1, scale=c2.scale)
        if need[4]:
            db2 = hip_ops.channel_sum(g2)
        adj2 = _cached_pack(c2, "adj", lambda: hip_ops.PackedConv(hip_ops.pack_weight(w2, adjoint=True, scale=c2.scale), 1, ci, co, 3, 3, 1, (1,), (1,)))
        dy1b = hip_ops.conv_transpose2d_s2_into(g2, adj2, y1b.shape[2:])
        dy1 = _blur_adjoint(dy1b, blur2, y1.shape[2:])
        g1 = hip_ops.fused_bias_act(dy1, dy1.new_empty(0), y1, 3, 1, 0.2, 2 ** 0.5)
        if need[1] and not no_w:
            dw1 = hip_ops.conv2d_wgrad(x, g1, tuple(w1.shape), 1, c1.padding, 1, 1, scale=c1.scale)
        if need[2]:
            db1 = hip_ops.channel_sum(g1)
        if need[0]:
            adj1 = _cached_pack(c1, "adj", lambda: hip_ops.PackedConv(
                hip_ops.pack_weight(w1, adjoint=True, flip=True, scale=c1.scale), 1, ci, ci, 3, 3, 1, (1,), (3 - 1 - c1.padding,)))
            dx = hip_ops.conv2d_packed(g1, adj1, res1=dx_skip)   # d/dx of the main path + the skip path's gradient in the epilogue
        return dx, dw1, db1, dw2, db2, dws, None


class ConvLayer(nn.Sequential):
    """[Blur,] EqualConv2d [, FusedLeakyReLU] with the reference's child indices (models/RestoreNet.py:1137-1179)."""

    def __init__(self, in_channel, out_channel, kernel_size, downsample=False, blur_kernel=(1, 3, 3, 1), bias=True, activate=True):
        layers = []
        if downsample:
            p = (len(blur_kernel) - 2) + (kernel_size - 1)
            layers.append(Blur(list(blur_kernel), pad=((p + 1) // 2, p // 2)))
        layers.append(EqualConv2d(in_channel, out_channel, kernel_size, padding=0 if downsample else kernel_size // 2,
                                  stride=2 if downsample else 1, bias=bias and not activate))
        if activate:
            layers.append(LeakyBias(out_channel))
        super().__init__(*layers)

    def forward(self, x, gain=2 ** 0.5):
        mods = list(self)
        fused = (_FIRST_ORDER and torch.is_grad_enabled() and isinstance(mods[-1], LeakyBias) and mods[-2].bias is None
                 and (mods[-2].stride == 1 or (mods[-2].weight.shape[2] == 3 and mods[-2].padding == 0)))
        for m in mods:
            if isinstance(m, Blur):
                x = upfirdn2d(x, m.kernel, pad=m.pad)
            elif isinstance(m, EqualConv2d):
                if fused:
                    return _ConvLrelu.apply(x, m.weight, mods[-1].bias, m, gain)
                x = conv2d_gradfix.conv2d(x, m.weight * m.scale, bias=m.bias, stride=m.stride, padding=m.padding)
            else:
                x = fused_leaky_relu(x, m.bias)
        return x


class ResBlock(nn.Module):
    def __init__(self, in_channel, out_channel, blur_kernel=(1, 3, 3, 1)):
        super().__init__()
        self.conv1 = ConvLayer(in_channel, in_channel, 3)
        self.conv2 = ConvLayer(in_channel, out_channel, 3, downsample=True, blur_kernel=blur_kernel)
        self.skip = ConvLayer(in_channel, out_channel, 1, downsample=True, blur_kernel=blur_kernel, activate=False, bias=False)

    def forward(self, x):
        if _FIRST_ORDER and torch.is_grad_enabled():
            # first-order passes: the 1 / sqrt 2 rides in conv2's activation gain and the skip's weight scale, the sum in the skip's epilogue
            if hip_ops.RESBLOCK_ONE_NODE and x.is_cuda:
                return _ResBlockFO.apply(x, self.conv1[0].weight, self.conv1[1].bias, self.conv2[1].weight, self.conv2[2].bias,
                                         self.skip[1].weight, self)
            main = self.conv2(self.conv1(x), gain=1.0)
            blur, conv = self.skip[0], self.skip[1]
            return _SkipConvAdd.apply(upfirdn2d(x, blur.kernel, pad=blur.pad), conv.weight, main, conv)
        return (self.conv2(self.conv1(x)) + self.skip(x)) / math.sqrt(2)


class Discriminator(nn.Module):
    def __init__(self, size, input_channel=3, channel_multiplier=2, blur_kernel=(1, 3, 3, 1)):
        super().__init__()
        cm = channel_multiplier
        channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * cm, 128: 128 * cm, 256: 64 * cm, 512: 32 * cm, 1024: 16 * cm}
        self.encoder_input_convs = ConvLayer(input_channel, channels[size], 1)
        self.log_size = int(math.log(size, 2))
        in_channel = channels[size]
        self.encoder_convs = nn.ModuleList()
        for i in range(self.log_size, 2, -1):
            out_channel = channels[2 ** (i - 1)]
            self.encoder_convs.append(ResBlock(in_channel, out_channel, blur_kernel))
            in_channel = out_channel
        self.stddev_group, self.stddev_feat = 4, 1
        self.final_conv = ConvLayer(in_channel + 1, channels[4], 3)
        self.final_linear = nn.Sequential(EqualLinear(channels[4] * 4 * 4, channels[4], activation="fused_lrelu"),
                                          EqualLinear(channels[4], 1))

    def forward(self, x):
        out = self.encoder_input_convs(x.contiguous())
        for blk in self.encoder_convs:
            out = blk(out)
        batch, channel, height, width = out.shape
        group = min(batch, self.stddev_group)                    # minibatch standard deviation (models/RestoreNet.py:1249-1256)
        sd = out.view(group, -1, self.stddev_feat, channel // self.stddev_feat, height, width)
        sd = torch.sqrt(sd.var(0, unbiased=False) + 1e-8).mean([2, 3, 4], keepdims=True).squeeze(2)
        out = torch.cat([out, sd.repeat(group, 1, height, width)], 1)
        out = self.final_conv(out)
        out = equal_linear(out.view(batch, -1), self.final_linear[0], twice_differentiable=True)
        return equal_linear(out, self.final_linear[1], twice_differentiable=True)


# ---------------------------------------------------------------------------------------------------- adversarial losses
def d_logistic_loss(real_pred, fake_pred):
    """restoration_train.py:60-64"""
    return F.softplus(-real_pred).mean() + F.softplus(fake_pred).mean()


def d_r1_loss(real_pred, real_img):
    """restoration_train.py:66-73: the input gradient is taken without weight gradients and kept differentiable."""
    with conv2d_gradfix.no_weight_gradients():
        grad_real, = torch.autograd.grad(outputs=real_pred.sum(), inputs=real_img, create_graph=True)
    return grad_real.pow(2).reshape(grad_real.shape[0], -1).sum(1).mean()


def g_nonsaturating_loss(fake_pred):
    """restoration_train.py:76-79"""
    return F.softplus(-fake_pred).mean()


def accumulate(model1, model2, decay=0.999):
    """EMA of the generator (restoration_train.py:46-51)."""
    par1, par2 = dict(model1.named_parameters()), dict(model2.named_parameters())
    with torch.no_grad():   # two multi-tensor launches instead of two per parameter
        a, b = [par1[k] for k in par1], [par2[k].detach() for k in par1]
        torch._foreach_mul_(a, decay)
        torch._foreach_add_(a, b, alpha=1 - decay)
