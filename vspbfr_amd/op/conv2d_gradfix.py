"""conv2d_gradfix: signature mirror of reference op/conv2d_gradfix.py:22-92 (which, on every torch other than 1.7/1.8, is a
pass-through to F.conv2d / F.conv_transpose2d, :78-92, i.e. plain autograd) over the gfx950 kernels.

Forward: the implicit-GEMM / Winograd kernels; `groups > 1` (the reference's `groups=batch` modulated form,
models/RestoreNet.py:373-383) is ONE launch through the kernel's true-group mode.  With autograd enabled the two functions are
differentiable in input, weight and bias, as the training step needs (restoration_train.py:153-255):
    data gradient    stride 1: the forward kernel with the flipped, channel-transposed weight; stride 2 (padding 0): the one-pass
                     transposed conv (3x3) or the 1x1 adjoint on the even pixels; transposed conv: the stride-2 conv
    weight gradient  vsp_conv2d_wgrad_f32 (skipped inside `no_weight_gradients()`, reference :12-19)
Both are written over these same public functions, so a backward pass run with create_graph=True can be differentiated again
(the reference's double backward, :104-227; the R1 penalty of the discriminator needs it).
Under torch.no_grad() (the restoration path) they are plain calls."""
import contextlib

import torch
from torch.autograd import Function

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


def _conv_fwd(x, weight, bias, stride, padding, dilation, groups):
    cout, cg, kh, kw = weight.shape
    if x.shape[1] != cg * groups or cout % groups:
        raise RuntimeError(f"conv2d_gradfix.conv2d: input has {x.shape[1]} channels, weight expects {cg} x {groups} groups")
    if groups == 1:
        return hip_ops.conv2d(x.contiguous(), weight.contiguous(), bias, stride, padding, dilation)
    pc = hip_ops.PackedConv(hip_ops.pack_weight(weight.contiguous(), groups), groups, cout // groups, cg, kh, kw, stride, (dilation,),
                            (padding,), x_group_stride=cg)
    return hip_ops.conv2d_packed(x.contiguous(), pc, ch_bias=None if bias is None else bias.contiguous())


def _convT_fwd(x, weight, groups):
    """conv_transpose2d(stride 2, padding 0), 3x3; weight (G*Cin_g, Cout_g, 3, 3)."""
    cg = x.shape[1] // groups
    if groups == 1:   # the one-pass transposed kernel (all four sub-pixel phases from one staged patch), whole batch in one launch
        pc = hip_ops.PackedConv(hip_ops.pack_weight(weight, adjoint=True), 1, weight.shape[1], cg, 3, 3, 1, (1,), (1,))
        return hip_ops.conv_transpose2d_s2_fused(x.contiguous(), pc)
    outs = []
    for g in range(groups):
        w_io = weight[g * cg:(g + 1) * cg]  # (Cin_g, Cout_g, 3, 3)
        phases = hip_ops.pack_transposed_s2(w_io.transpose(0, 1).contiguous())
        outs.append(hip_ops.conv_transpose2d_s2(x[:, g * cg:(g + 1) * cg].contiguous(), phases))
    return outs[0] if groups == 1 else torch.cat(outs, dim=1)


def _dgrad(g, weight, x_shape, stride, padding, dilation, groups):
    """Data gradient of conv2d as a composition of the PUBLIC (differentiable) entry points, so that a backward pass run with
    create_graph=True (the R1 penalty, restoration_train.py:66-73) can be differentiated again: it is linear in g and in weight."""
    cout, cg, kh, kw = weight.shape
    if stride == 1:
        # adjoint of a stride-1 correlation: correlation of g with the flipped kernel, channels exchanged inside each group
        if (groups == 1 and not torch.is_grad_enabled() and kh == 1 and kw == 1 and padding == 0 and cout % 16 == 0
                and x_shape[2] * x_shape[3] <= hip_ops.CONV1X1_SMALL_MAX_P):
            # 1x1 on a small map (the bottlenecks of the identity network): the data gradient is the GEMM W^T g
            key = (weight._version, weight.data_ptr(), "1x1")
            cached = getattr(weight, "_vsp_adjoint", None)
            if cached is None or cached[0] != key:
                cached = (key, weight.detach().view(cout, cg).t().contiguous())
                try:
                    weight._vsp_adjoint = cached
                except (AttributeError, RuntimeError):
                    pass
            return hip_ops.conv1x1_small(g.contiguous(), cached[1])
        if groups == 1 and not torch.is_grad_enabled():
            # first-order pass: the adjoint weight is packed by one launch, once per weight version (a frozen loss network keeps
            # it for the whole run; the product `weight * scale` of an equalised layer is a temporary and takes its packing with it)
            key = (weight._version, weight.data_ptr(), padding, dilation)
            cached = getattr(weight, "_vsp_adjoint", None)
            if cached is None or cached[0] != key:
                cached = (key, hip_ops.PackedConv(hip_ops.pack_weight(weight, adjoint=True, flip=True), 1, cg, cout, kh, kw, 1,
                                                  (dilation,), (dilation * (kh - 1) - padding,)))
                try:
                    weight._vsp_adjoint = cached
                except (AttributeError, RuntimeError):
                    pass
            return hip_ops.conv2d_packed(g.contiguous(), cached[1])
        if not weight.requires_grad and not torch.is_grad_enabled():
            # frozen grouped weight, first-order pass: the adjoint weight is built -- and packed -- once per version
            key = (weight._version, weight.data_ptr(), groups)
            cached = getattr(weight, "_vsp_adjoint", None)
            if cached is None or cached[0] != key:
                cached = (key, weight.reshape(groups, cout // groups, cg, kh, kw).transpose(1, 2).flip(3, 4)
                          .reshape(groups * cg, cout // groups, kh, kw).contiguous())
                try:
                    weight._vsp_adjoint = cached
                except (AttributeError, RuntimeError):
                    pass
            return conv2d(g, cached[1], None, 1, dilation * (kh - 1) - padding, dilation, groups)
        wt = weight.reshape(groups, cout // groups, cg, kh, kw).transpose(1, 2).flip(3, 4).reshape(groups * cg, cout // groups, kh, kw)
        return conv2d(g, wt, None, 1, dilation * (kh - 1) - padding, dilation, groups)
    if stride == 2 and padding in (0, 1) and dilation == 1 and (kh, kw) == (3, 3):
        # adjoint of the stride-2 conv = conv_transpose2d(g, W, stride 2): (2 OH + 1)^2 rows of the PADDED input, zero rows beyond
        # when H + 2 padding is even; padding 1 (the ResNet bottleneck of the identity loss) crops the border back off
        if groups == 1 and padding == 0 and not torch.is_grad_enabled():
            # first-order pass: the one-pass transposed kernel writes straight into the (H, W) gradient (no padded copy)
            # conv_transpose2d reads (Cout, Cin, 3, 3) as (in, out): the adjoint packing of the weight as it lies
            pc = hip_ops.PackedConv(hip_ops.pack_weight(weight, adjoint=True), 1, cg, cout, 3, 3, 1, (1,), (1,))
            return hip_ops.conv_transpose2d_s2_into(g.contiguous(), pc, x_shape[2:])
        dx = conv_transpose2d(g, weight, stride=2, padding=0, groups=groups)
        ph, pw = x_shape[2] + 2 * padding - dx.shape[2], x_shape[3] + 2 * padding - dx.shape[3]
        if ph < 0 or pw < 0 or ph > 1 or pw > 1:
            raise RuntimeError("conv2d_gradfix: stride-2 data gradient needs H + 2 padding in {2 OH + 1, 2 OH + 2}")
        if ph or pw:
            dx = torch.nn.functional.pad(dx, (0, pw, 0, ph))
        return dx[:, :, padding:padding + x_shape[2], padding:padding + x_shape[3]] if padding else dx
    if stride == 2 and padding == 0 and (kh, kw) == (1, 1):
        # 1x1, stride 2 (the skip branch of the discriminator's ResBlock): the 1x1 adjoint lands on the even pixels
        wt = weight.reshape(groups, cout // groups, cg, 1, 1).transpose(1, 2).reshape(groups * cg, cout // groups, 1, 1)
        d = conv2d(g, wt, None, 1, 0, 1, groups)
        dx = d.new_zeros(x_shape)
        dx[:, :, 0:2 * d.shape[2]:2, 0:2 * d.shape[3]:2] = d
        return dx
    raise RuntimeError("conv2d_gradfix: the data gradient is implemented for stride 1, and for stride 2 with 3x3 (padding 0 / 1) and "
                       "1x1 (padding 0) kernels (the forms of restoration_train.py and its loss networks)")


class _Wgrad(Function):
    """dW = wgrad(x, g) of conv2d -- bilinear in (x, g), so its own backward is a conv (w.r.t. g) and a data gradient (w.r.t. x)."""

    @staticmethod
    def forward(ctx, x, g, weight_shape, stride, padding, dilation, groups):
        ctx.save_for_backward(x, g)
        ctx.cfg = (tuple(weight_shape), stride, padding, dilation, groups)
        return hip_ops.conv2d_wgrad(x.detach().contiguous(), g.detach().contiguous(), weight_shape, stride, padding, dilation, groups)

    @staticmethod
    def backward(ctx, ggw):
        x, g = ctx.saved_tensors
        _, stride, padding, dilation, groups = ctx.cfg
        dx = dg = None
        if ctx.needs_input_grad[0]:
            dx = _dgrad(g, ggw, x.shape, stride, padding, dilation, groups)
        if ctx.needs_input_grad[1]:
            dg = conv2d(x, ggw, None, stride, padding, dilation, groups)
        return dx, dg, None, None, None, None, None


class _Conv2d(Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, dilation, groups):
        ctx.save_for_backward(x, weight)
        ctx.cfg = (stride, padding, dilation, groups, bias is not None)
        return _conv_fwd(x, weight, bias, stride, padding, dilation, groups)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        stride, padding, dilation, groups, has_bias = ctx.cfg
        g = g.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _dgrad(g, weight, x.shape, stride, padding, dilation, groups)
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            dw = _Wgrad.apply(x, g, weight.shape, stride, padding, dilation, groups)
        if has_bias and ctx.needs_input_grad[2]:
            db = g.sum((0, 2, 3)) if torch.is_grad_enabled() else hip_ops.channel_sum(g)
        return dx, dw, db, None, None, None, None


class _WgradT(Function):
    """dW of conv_transpose2d(x, W, stride 2): the stride-2 conv weight gradient with the roles of x and g exchanged."""

    @staticmethod
    def forward(ctx, x, g, weight_shape, groups):
        ctx.save_for_backward(x, g)
        ctx.cfg = (tuple(weight_shape), groups)
        return hip_ops.conv2d_wgrad(g.detach().contiguous(), x.detach().contiguous(), weight_shape, 2, 0, 1, groups)

    @staticmethod
    def backward(ctx, ggw):
        x, g = ctx.saved_tensors
        _, groups = ctx.cfg
        dx = dg = None
        if ctx.needs_input_grad[0]:
            dx = conv2d(g, ggw, None, 2, 0, 1, groups)
        if ctx.needs_input_grad[1]:
            dg = conv_transpose2d(x, ggw, stride=2, padding=0, groups=groups)
        return dx, dg, None, None


class _ConvTranspose2d(Function):
    @staticmethod
    def forward(ctx, x, weight, groups):
        ctx.save_for_backward(x, weight)
        ctx.groups = groups
        return _convT_fwd(x, weight, groups)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        groups = ctx.groups
        g = g.contiguous()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            # x[ci, m, n] meets g[co, 2m + ky, 2n + kx] through W[ci][co][ky][kx]: a stride-2 conv of g with W read as (out = ci, in = co)
            dx = conv2d(g, weight, None, 2, 0, 1, groups)
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            # the same sum with the roles exchanged: "input" = g (Cout_g channels per group), "output gradient" = x
            dw = _WgradT.apply(x, g, weight.shape, groups)
        return dx, dw, None


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    (sy, sx), (py, px), (dy, dx) = _pair(stride), _pair(padding), _pair(dilation)
    if sy != sx or py != px or dy != dx:
        raise RuntimeError("conv2d_gradfix.conv2d: only square stride/padding/dilation are supported")
    if torch.is_grad_enabled() and (input.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
        return _Conv2d.apply(input, weight, bias, sy, py, dy, groups)
    return _conv_fwd(input, weight, bias, sy, py, dy, groups)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    (sy, sx) = _pair(stride)
    if (sy, sx) != (2, 2) or _pair(padding) != (0, 0) or _pair(output_padding) != (0, 0) or _pair(dilation) != (1, 1) \
            or tuple(weight.shape[2:]) != (3, 3) or bias is not None:
        raise RuntimeError("conv2d_gradfix.conv_transpose2d: the restoration path only uses stride=2, padding=0, 3x3, "
                           "no bias (models/RestoreNet.py:530-532); other forms are not implemented")
    if input.shape[1] != weight.shape[0]:
        raise RuntimeError("conv2d_gradfix.conv_transpose2d: weight is (G*Cin_g, Cout_g, 3, 3)")
    if torch.is_grad_enabled() and (input.requires_grad or weight.requires_grad):
        return _ConvTranspose2d.apply(input, weight, groups)
    return _convT_fwd(input, weight, groups)
