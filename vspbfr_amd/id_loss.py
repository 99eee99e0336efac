"""ArcFace identity loss of the training step (BASELINE configs[4]; reference restoration_train.py:116-117, 242-245 and
Loss/id_loss.py:7-46) over the gfx950 operators.

    IDLoss.forward(target_img, source_img):  z = normalize(Z(interpolate(img, 112, bilinear)));  loss = L1(1, <z_src.detach(), z_out>)
    Z = torchvision.models.resnet101(num_classes=256).eval(), frozen (Loss/id_loss.py:13-15); state dict = torchvision's
    (`conv1.weight`, `bn1.*`, `layer{1..4}.{i}.conv{1,2,3}.weight`, `.bn{1,2,3}.*`, `.downsample.{0,1}.*`, `fc.*`), so the checkpoint
    `--arcface_path` names loads with strict=True.

The network runs in eval mode only: every BatchNorm is folded into its convolution (w' = w * gamma / sqrt(var + eps),
b' = beta - mean * gamma / sqrt(var + eps); cached until a parameter changes), so a layer is conv + bias (+ ReLU) =
`conv2d_gradfix.conv2d` + `fused_leaky_relu(x, b', 0, 1)`.  The 7x7 stride-2 stem is evaluated as a 4x4 stride-1 convolution over the
four sub-pixel phases of the padded input (12 channels), which keeps its data gradient on the stride-1 forward kernel; the 3x3
stride-2 bottleneck convs use the padding-1 form of the stride-2 data gradient; max-pooling `vsp_maxpool2d_f32`; the 112x112
resize `vsp_resize_bilinear_f32` with its adjoint `vsp_resize_bilinear_bwd_f32`.  Only the data gradient exists (frozen network)."""
import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import hip_ops
from .lpips import max_pool2d
from .op import conv2d_gradfix, fused_leaky_relu


class _Resize(Function):
    @staticmethod
    def forward(ctx, x, size):
        ctx.in_size = tuple(x.shape[-2:])
        return hip_ops.resize_bilinear(x.contiguous(), size)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return hip_ops.resize_bilinear_bwd(g.contiguous(), ctx.in_size), None


def interpolate_bilinear(x, size):
    size = (size, size) if isinstance(size, int) else tuple(size)
    if torch.is_grad_enabled() and x.requires_grad:
        return _Resize.apply(x, size)
    return hip_ops.resize_bilinear(x.contiguous(), size)


class _BN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight, self.bias = nn.Parameter(torch.ones(c)), nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
        self.eps = 1e-5


class _ConvW(nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k).normal_(0, (2.0 / (cin * k * k)) ** 0.5))


def _folded(conv, bn):
    """(w', b') of conv followed by eval-mode BatchNorm; cached on the conv module until any of the five tensors changes."""
    ver = (conv.weight._version, bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           conv.weight.data_ptr())
    cache = getattr(conv, "_fold_cache", None)
    if cache is None or cache[0] != ver:
        with torch.no_grad():
            s = bn.weight * torch.rsqrt(bn.running_var + bn.eps)
            cache = (ver, (conv.weight * s.view(-1, 1, 1, 1)).contiguous(), (bn.bias - bn.running_mean * s).contiguous())
        conv._fold_cache = cache
    return cache[1], cache[2]


class _ConvBiasAct(Function):
    """conv (folded BatchNorm) + bias (+ ReLU) of the FROZEN network as one launch with a data gradient only: forward = the conv
    kernel's fused epilogue, backward = the ReLU mask from y (fused_bias_act act=3 grad=1 with slope 0) and the data-gradient conv
    (adjoint weight cached on the folded tensor).  Halves the launches of the 104-layer network, which is launch-bound at 112x112."""

    @staticmethod
    def forward(ctx, x, w, b, stride, padding, relu):
        x = x.contiguous()
        if relu:
            y = hip_ops.conv2d(x, w, None, stride, padding, 1, act2=1, bias2=b, slope2=0.0, gain2=1.0)
        else:
            y = hip_ops.conv2d(x, w, b, stride, padding, 1)
        ctx.cfg = (tuple(x.shape), stride, padding, relu)
        ctx.w = w
        ctx.save_for_backward(y if relu else None)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        x_shape, stride, padding, relu = ctx.cfg
        g = g.contiguous()
        if relu:
            g = hip_ops.fused_bias_act(g, g.new_empty(0), y, 3, 1, 0.0, 1.0)
        return conv2d_gradfix._dgrad(g, ctx.w, x_shape, stride, padding, 1, 1), None, None, None, None, None


def _conv_bn(x, conv, bn, stride=1, padding=0, relu=True):
    w, b = _folded(conv, bn)
    if torch.is_grad_enabled() and x.requires_grad:
        return _ConvBiasAct.apply(x, w, b, stride, padding, relu)
    if relu:
        return hip_ops.conv2d(x.contiguous(), w, None, stride, padding, 1, act2=1, bias2=b, slope2=0.0, gain2=1.0)
    return hip_ops.conv2d(x.contiguous(), w, b, stride, padding, 1)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=False):
        super().__init__()
        self.conv1, self.bn1 = _ConvW(inplanes, planes, 1), _BN(planes)
        self.conv2, self.bn2 = _ConvW(planes, planes, 3), _BN(planes)
        self.conv3, self.bn3 = _ConvW(planes, planes * 4, 1), _BN(planes * 4)
        self.stride = stride
        self.downsample = None
        if downsample:
            self.downsample = nn.Sequential()
            self.downsample.add_module("0", _ConvW(inplanes, planes * 4, 1))
            self.downsample.add_module("1", _BN(planes * 4))

    def forward(self, x):
        out = _conv_bn(x, self.conv1, self.bn1)
        out = _conv_bn(out, self.conv2, self.bn2, self.stride, 1)
        out = _conv_bn(out, self.conv3, self.bn3, relu=False)
        idt = x if self.downsample is None else _conv_bn(x, getattr(self.downsample, "0"), getattr(self.downsample, "1"), self.stride, 0,
                                                         relu=False)
        return fused_leaky_relu(out + idt, None, 0.0, 1.0)


class ResNet101(nn.Module):
    """torchvision 0.13 `resnet101(num_classes=...)` (v1.5: the 3x3 conv of a bottleneck carries the stride), eval mode."""

    def __init__(self, num_classes=256, layers=(3, 4, 23, 3)):
        super().__init__()
        self.conv1, self.bn1 = _ConvW(3, 64, 7), _BN(64)
        inplanes = 64
        for li, (planes, n, stride) in enumerate(zip((64, 128, 256, 512), layers, (1, 2, 2, 2))):
            blocks = [Bottleneck(inplanes, planes, stride, downsample=(stride != 1 or inplanes != planes * 4))]
            inplanes = planes * 4
            blocks += [Bottleneck(inplanes, planes) for _ in range(1, n)]
            setattr(self, f"layer{li + 1}", nn.Sequential(*blocks))
        self.fc = nn.Linear(512 * 4, num_classes)

    def _stem(self, x):
        # conv 7x7, stride 2, padding 3  ==  conv 4x4, stride 1 over the 2x2 sub-pixel phases of the padded input:
        #   y[i, j] = sum_{u, v < 7} w[u, v] xp[2 i + u, 2 j + v],  u = 2 a + r  ->  phase plane r of xp at row i + a, tap a of 4 (w[7] = 0)
        w, b = _folded(self.conv1, self.bn1)
        cache = getattr(self, "_stem_cache", None)
        if cache is None or cache[0] is not w:
            w8 = F.pad(w, (0, 1, 0, 1))                                                     # (64, 3, 8, 8)
            w4 = w8.reshape(64, 3, 4, 2, 4, 2).permute(0, 1, 3, 5, 2, 4).reshape(64, 12, 4, 4).contiguous()
            cache = (w, w4)
            self._stem_cache = cache
        w4 = cache[1]
        B, C, H, W = x.shape
        if H % 2 or W % 2:
            raise RuntimeError("id_loss: the stem expects an even input size (112 in Loss/id_loss.py)")
        xp = F.pad(x, (3, 3, 3, 3))
        Hp, Wp = H + 6, W + 6
        ph = xp.reshape(B, C, Hp // 2, 2, Wp // 2, 2).permute(0, 1, 3, 5, 2, 4).reshape(B, C * 4, Hp // 2, Wp // 2).contiguous()
        if torch.is_grad_enabled() and x.requires_grad:
            return _ConvBiasAct.apply(ph, w4, b, 1, 0, True)
        return hip_ops.conv2d(ph, w4, None, 1, 0, 1, act2=1, bias2=b, slope2=0.0, gain2=1.0)

    def forward(self, x):
        if x.device.type != "cuda":
            raise RuntimeError("vspbfr_amd.id_loss: inputs must be CUDA (HIP) tensors; there is no CPU path")
        if self.training:
            raise RuntimeError("vspbfr_amd.id_loss: the identity network runs in eval mode only (Loss/id_loss.py:13)")
        h = max_pool2d(self._stem(x), 3, 2, 1)
        for li in range(4):
            h = getattr(self, f"layer{li + 1}")(h)
        return F.linear(h.mean([2, 3]), self.fc.weight, self.fc.bias)


class IDLoss(nn.Module):
    """`Loss.id_loss.IDLoss(model_path)`; model_path: a torchvision resnet101(num_classes=256) state dict (file or dict), or None."""

    def __init__(self, model_path=None, device="cuda"):
        super().__init__()
        self.Z = ResNet101(num_classes=256).eval()
        self.Z.requires_grad_(False)
        if model_path is not None:
            sd = torch.load(model_path, map_location="cpu") if isinstance(model_path, str) else model_path
            self.Z.load_state_dict(sd)
        self.Z.to(device)

    @staticmethod
    def id_loss(z_id_X, z_id_Y):
        inner = (z_id_X * z_id_Y).sum(1)                      # bmm of (B,1,D) x (B,D,1), squeezed
        return (1.0 - inner).abs().mean()                     # nn.L1Loss()(ones, inner)

    def get_id(self, target_img, weight_map=None):
        if weight_map is not None:
            raise RuntimeError("vspbfr_amd.id_loss: weight_map (SVGL.ada_piexls) is not on the path of restoration_train.py")
        return F.normalize(self.Z(interpolate_bilinear(target_img, 112)))

    def forward(self, target_img, source_img, weight_map=None):
        if weight_map is not None:
            raise RuntimeError("vspbfr_amd.id_loss: weight_map (SVGL.ada_piexls) is not on the path of restoration_train.py")
        if torch.is_grad_enabled() and target_img.requires_grad and source_img.shape == target_img.shape:
            # one pass over [source, target] (eval-mode BatchNorm: no cross-sample term): half the launches of a launch-bound network;
            # the source half is detached on both ends, so it carries no gradient
            B = target_img.shape[0]
            z = self.get_id(torch.cat([source_img.detach(), target_img], 0))
            return self.id_loss(z[:B].detach(), z[B:])
        with torch.no_grad():
            z_id = self.get_id(source_img)
        return self.id_loss(z_id, self.get_id(target_img, weight_map))
