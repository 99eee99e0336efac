"""LPIPS-VGG perceptual loss of the training step (BASELINE configs[4]; reference restoration_train.py:143, 236-239:
`my_lpips.PerceptualLoss(model="net-lin", net="vgg")`, `percept_loss(restored, real).sum() * 0.5`) over the gfx950 operators.

    PerceptualLoss.forward(pred, target)        my_lpips/__init__.py:27-42   -> model.forward(target, pred): in0 = target, in1 = pred
    PNetLin.forward                             my_lpips/networks_basic.py:65-96
        ScalingLayer (x - shift) / scale        :98-105
        vgg16 slices relu1_2 ... relu5_3        my_lpips/pretrained_networks.py:97-135 (torchvision 0.13 vgg16().features[0:30])
        normalize_tensor, (f0 - f1)^2, lin_k (1x1 conv, no bias; Dropout is the identity in eval), spatial average, sum over levels

State dict = the reference's (`net.slice{1..5}.{0,2,5,7,10,12,14,17,19,21,24,26,28}.{weight,bias}`, `lin{0..4}.model.1.weight`,
`scaling_layer.{shift,scale}`), so `my_lpips/weights/v0.1/vgg.pth` loads with strict=False exactly as dist_model.py:69 does, and a
torchvision VGG16 checkpoint loads into `net` through `load_vgg16_features`.

Convolutions run on conv2d_gradfix (Winograd / implicit-GEMM kernels; data gradient on the forward kernels, no weight gradient: the
network is frozen), bias + ReLU is `fused_leaky_relu(x, b, 0, 1)`, max-pooling `vsp_maxpool2d_f32`, and everything after the
features of a level -- two channel norms, the weighted squared difference, the spatial mean -- is ONE launch
(`vsp_lpips_layer_f32`, gradient `vsp_lpips_layer_bwd_f32`)."""
import torch
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import hip_ops
from .op import conv2d_gradfix, fused_leaky_relu

# torchvision vgg16 "D": index in `features` -> out channels; "M" = MaxPool2d(2, 2) (ReLU follows every conv)
_VGG16 = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
_SLICE_ENDS = (4, 9, 16, 23, 30)      # features[0:4], [4:9], [9:16], [16:23], [23:30]  (the last MaxPool is not used)
CHNS = (64, 128, 256, 512, 512)


class _MaxPool(Function):
    @staticmethod
    def forward(ctx, x, k, s, p):
        ctx.save_for_backward(x)
        ctx.cfg = (k, s, p)
        return hip_ops.maxpool2d(x.contiguous(), k, s, p)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return hip_ops.maxpool2d_bwd(g.contiguous(), x.contiguous(), *ctx.cfg), None, None, None


def max_pool2d(x, k, s, p=0):
    if torch.is_grad_enabled() and x.requires_grad:
        return _MaxPool.apply(x, k, s, p)
    return hip_ops.maxpool2d(x.contiguous(), k, s, p)


class _LayerDistance(Function):
    """(B,) level distance; symmetric in (f0, f1), so either gradient is the same kernel with the roles exchanged."""

    @staticmethod
    def forward(ctx, f0, f1, w):
        ctx.save_for_backward(f0, f1, w)
        return hip_ops.lpips_layer(f0.contiguous(), f1.contiguous(), w.contiguous())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        f0, f1, w = ctx.saved_tensors
        g = g.contiguous()
        d0 = hip_ops.lpips_layer_bwd(g, f1.contiguous(), f0.contiguous(), w.contiguous()) if ctx.needs_input_grad[0] else None
        d1 = hip_ops.lpips_layer_bwd(g, f0.contiguous(), f1.contiguous(), w.contiguous()) if ctx.needs_input_grad[1] else None
        return d0, d1, None


class _Conv(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 3).normal_(0, (2.0 / (9 * cin)) ** 0.5))
        self.bias = nn.Parameter(torch.zeros(cout))


class VGG16Slices(nn.Module):
    """`pretrained_networks.vgg16`: five Sequential slices holding the convs under their torchvision `features` indices."""

    def __init__(self, requires_grad=False):
        super().__init__()
        self.N_slices = 5
        self.plan = []                      # per slice: list of ("conv", module) / ("pool",)
        idx, cin, si = 0, 3, 0
        slices = [nn.Sequential() for _ in range(5)]
        plan = [[] for _ in range(5)]
        for v in _VGG16:
            if idx >= _SLICE_ENDS[-1]:
                break
            while idx >= _SLICE_ENDS[si]:
                si += 1
            if v == "M":
                plan[si].append(("pool", None))
                idx += 1
            else:
                conv = _Conv(cin, v)
                slices[si].add_module(str(idx), conv)
                plan[si].append(("conv", str(idx)))
                cin = v
                idx += 2
        for i, s in enumerate(slices):
            setattr(self, f"slice{i + 1}", s)
        self.plan = plan
        if not requires_grad:
            for p in self.parameters():
                p.requires_grad = False

    def forward(self, x):
        outs = []
        h = x
        for i in range(5):
            sl = getattr(self, f"slice{i + 1}")
            for kind, name in self.plan[i]:
                if kind == "pool":
                    h = max_pool2d(h, 2, 2)
                else:
                    c = getattr(sl, name)
                    if torch.is_grad_enabled() and c.weight.requires_grad:
                        h = fused_leaky_relu(conv2d_gradfix.conv2d(h, c.weight, padding=1), c.bias, 0.0, 1.0)
                    elif torch.is_grad_enabled() and h.requires_grad:
                        # frozen weights, gradient w.r.t. the image only (the loss network of restoration_train.py:237-241):
                        # bias + ReLU in the conv epilogue, backward = ReLU mask from y + data-gradient conv
                        from .id_loss import _ConvBiasAct
                        h = _ConvBiasAct.apply(h, c.weight, c.bias, 1, 1, True)
                    else:   # frozen branch: bias + ReLU in the conv epilogue
                        h = hip_ops.conv2d(h.contiguous(), c.weight, None, 1, 1, 1, act2=1, bias2=c.bias, slope2=0.0, gain2=1.0)
            outs.append(h)
        return outs


class ScalingLayer(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("shift", torch.tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer("scale", torch.tensor([.458, .448, .450])[None, :, None, None])

    def forward(self, inp):
        return (inp - self.shift) / self.scale


class NetLinLayer(nn.Module):
    """`lin_k.model` = [Dropout, Conv2d(chn, 1, 1, bias=False)]: the conv sits at index 1 of the Sequential."""

    def __init__(self, chn_in):
        super().__init__()
        conv = nn.Module()
        conv.weight = nn.Parameter(torch.full((1, chn_in, 1, 1), 1.0 / chn_in))
        self.model = nn.Sequential()
        self.model.add_module("1", conv)

    @property
    def weight(self):
        return getattr(self.model, "1").weight


class PNetLin(nn.Module):
    def __init__(self, pnet_tune=False):
        super().__init__()
        self.scaling_layer = ScalingLayer()
        self.chns, self.L = list(CHNS), 5
        self.net = VGG16Slices(requires_grad=pnet_tune)
        for k, c in enumerate(CHNS):
            setattr(self, f"lin{k}", NetLinLayer(c))

    def forward(self, in0, in1, retPerLayer=False):
        if in0.device.type != "cuda":
            raise RuntimeError("vspbfr_amd.lpips: inputs must be CUDA (HIP) tensors; there is no CPU path")
        B = in0.shape[0]
        grad = torch.is_grad_enabled()
        if not (grad and (in0.requires_grad or in1.requires_grad)):
            with torch.no_grad():                                               # one pass for both images
                feats = self.net(self.scaling_layer(torch.cat([in0, in1], 0)))
            f0s, f1s = [f[:B] for f in feats], [f[B:] for f in feats]
        else:
            with torch.set_grad_enabled(in0.requires_grad):
                f0s = self.net(self.scaling_layer(in0))
            with torch.set_grad_enabled(in1.requires_grad):
                f1s = self.net(self.scaling_layer(in1))
        res = []
        for k in range(self.L):
            w = getattr(self, f"lin{k}").weight.reshape(-1)
            if torch.is_grad_enabled() and (f0s[k].requires_grad or f1s[k].requires_grad):
                d = _LayerDistance.apply(f0s[k], f1s[k], w)
            else:
                d = hip_ops.lpips_layer(f0s[k].contiguous(), f1s[k].contiguous(), w.contiguous())
            res.append(d.view(B, 1, 1, 1))
        val = res[0]
        for r in res[1:]:
            val = val + r
        return (val, res) if retPerLayer else val


class PerceptualLoss(nn.Module):
    """`my_lpips.PerceptualLoss(model="net-lin", net="vgg")`.  `lin_weights`: path of the reference's my_lpips/weights/v0.1/vgg.pth
    (loaded with strict=False as dist_model.py:69 does) or None (uniform 1/C); `vgg_weights`: a torchvision vgg16 state dict / path."""

    def __init__(self, model="net-lin", net="vgg", lin_weights=None, vgg_weights=None):
        super().__init__()
        if model != "net-lin" or net not in ("vgg", "vgg16"):
            raise RuntimeError("vspbfr_amd.lpips: only the configuration restoration_train.py uses is built (net-lin, vgg)")
        self.net = PNetLin()
        if lin_weights is not None:
            sd = torch.load(lin_weights, map_location="cpu") if isinstance(lin_weights, str) else lin_weights
            self.net.load_state_dict(sd, strict=False)
        if vgg_weights is not None:
            load_vgg16_features(self.net.net, vgg_weights)
        self.net.eval()
        self.net.requires_grad_(False)      # a frozen metric: the `lin` layers included (the reference leaves theirs trainable but never steps them)

    def forward(self, pred, target, normalize=False, weight_map=None):
        if weight_map is not None:
            raise RuntimeError("vspbfr_amd.lpips: weight_map (SVGL.ada_piexls) is not on the path of restoration_train.py")
        if normalize:
            target, pred = 2 * target - 1, 2 * pred - 1
        return self.net(target, pred)


def load_vgg16_features(slices, state):
    """Fill VGG16Slices from a torchvision vgg16 checkpoint (`features.<idx>.weight/bias`)."""
    sd = torch.load(state, map_location="cpu") if isinstance(state, str) else state
    own = slices.state_dict()
    for k in own:
        _, idx, leaf = k.split(".")
        own[k].copy_(sd[f"features.{idx}.{leaf}"])
