"""StyleGAN2-style building blocks shared by Restoration_net and the e4e prior decoder, executed on the gfx950 kernels.

Parameter/buffer names and shapes follow the reference's checkpoints so `load_state_dict(strict=True)`
works on the published files; the forward passes do NOT follow the reference's op sequence.  MI355X-first choices:

* modulate-input / demodulate-output: the per-sample weight tensor (B*Cout, Cin, 3, 3) of the reference's grouped conv
  (models/RestoreNet.py:373-416) is never built.  The style scales the input while the conv kernel stages its LDS patch,
  the demodulation coefficient rsqrt(scale^2 * sum_ci s^2 * sum_taps W^2 + 1e-8) is a (B, Cout) vector applied in the
  conv epilogue (same algebra as the reference's own `fused=False` branch, :343-370).
* weights are packed ONCE per device into the kernel layout [group][tap][ci][co] with the equalised-lr scale folded in.
* noise injection, bias, leaky-ReLU, residual adds and channel concatenation run inside the producing kernel's epilogue
  (conv) or the blur's epilogue (up-sampling convs): no standalone elementwise pass over a (B, C, H, W) tensor.
* a stride-2 transposed conv runs all four sub-pixel phases in one launch from one staged input patch (same MACs as
  conv_transpose2d), followed by the FIR blur whose epilogue carries noise, bias, leaky-ReLU and the residual adds.
Inference only (no autograd through the HIP ops).
"""
import ctypes
import math
import os

import torch
from torch import nn

from . import hip_ops as H
from ._lib import tune_env

SQRT2 = math.sqrt(2.0)
RATES = (1, 2, 4, 8)


def make_kernel(k):
    k = torch.tensor(k, dtype=torch.float32)
    if k.ndim == 1:
        k = k[None, :] * k[:, None]
    k /= k.sum()
    return k


class _Cached(nn.Module):
    """Device-side derived tensors (packed weights, folded constants), rebuilt when a source parameter changes."""

    def _derive(self, key, sources, fn):
        store = self.__dict__.setdefault("_derived", {})
        stamp = tuple((t.data_ptr(), t._version, t.device) for t in sources)
        hit = store.get(key)
        if hit is None or hit[0] != stamp:
            with torch.no_grad():
                hit = (stamp, fn())
            store[key] = hit
        return hit[1]

    def _derive_static(self, key, sources, fn):
        """_derive for a tensor whose ADDRESS is handed out (style-plan tables, captured graphs): refreshed in place when a source changes."""
        store = self.__dict__.setdefault("_derived", {})
        stamp = tuple((t.data_ptr(), t._version, t.device) for t in sources)
        hit = store.get(key)
        if hit is None or hit[0] != stamp:
            with torch.no_grad():
                new = fn()
                if hit is not None and hit[1].shape == new.shape and hit[1].device == new.device and hit[1].dtype == new.dtype:
                    new = hit[1].copy_(new)
                hit = (stamp, new)
            store[key] = hit
        return hit[1]


# ---- all style modulations of a network up front (vsp_style_plan_f32: two launches instead of ~2 per modulated layer)
STYLE_PLANS = tune_env("VSP_STYLE_PLAN", "1") != "0"
_STYLE_CTX = None


class _StyleLayerC(ctypes.Structure):   # vsp_style_layer (include/vspbfr_hip.h)
    _fields_ = [("w", ctypes.c_void_p), ("bias", ctypes.c_void_p), ("wsq", ctypes.c_void_p), ("mod", ctypes.c_void_p), ("demod", ctypes.c_void_p),
                ("src_off", ctypes.c_int64), ("cin", ctypes.c_int), ("cout", ctypes.c_int), ("alpha", ctypes.c_float),
                ("bias_scale", ctypes.c_float), ("wscale2", ctypes.c_float), ("pad_", ctypes.c_int)]


class StyleContext:
    """Per (network part, latent tensor): the first no-grad forward RECORDS which modulation layer reads which row view of `src`
    (the layer itself, its EqualLinear, whether it demodulates), the following ones evaluate all of them with vsp_style_plan_f32 before the
    first layer runs and hand the slices out.  A layer whose style is not the recorded row view of the current `src` falls back to its own
    two launches.

    The kernel reads weights, biases and squared-tap sums through LIVE pointers (the tap sums are refreshed in place when a weight changes:
    `_Cached._derive_static`, fetched from the layer at every `begin`), so a parameter UPDATE needs no rebuild and is never stale; only a
    moved pointer or another batch size builds a table.  Tables and output buffers are kept per batch size and never freed while the
    context lives: captured HIP graphs (pipeline.capture_graphs) hold their addresses."""
    MAX_RETIRED = 32

    def __init__(self):
        self.recorded = None     # list of (owner, eql, src_off, has_wsq, wscale) in call order
        self.plans = {}          # (B, rows, K, device) -> plan
        self.retired = []        # plans replaced because a pointer moved: kept alive (graphs may still replay them)

    def __deepcopy__(self, memo):   # a copied network (EMA copy) records for itself
        return StyleContext()

    @property
    def plan(self):              # the most recently used plan (tests)
        return self.__dict__.get("_last")

    def begin(self, src):
        self.src, self.hits, self.rec = src, None, None
        if not (STYLE_PLANS and src.is_cuda and src.dtype == torch.float32 and src.dim() == 3 and src.is_contiguous() and not torch.is_grad_enabled()):
            return
        B, K = src.shape[0], src.shape[2]
        if B > 16 or K < 512 or K % 256 or src.data_ptr() % 16:
            return
        if self.recorded is None:
            self.rec = []
            return
        wsqs = [owner._style_wsq() if has_wsq else None for owner, _e, _o, has_wsq, _s in self.recorded]
        ptrs = tuple((e[1].weight.data_ptr(), 0 if e[1].bias is None else e[1].bias.data_ptr(), 0 if q is None else q.data_ptr())
                     for e, q in zip(self.recorded, wsqs))
        key = (B, src.shape[1], K, src.device)
        plan = self.plans.get(key)
        if plan is None or plan["ptrs"] != ptrs:
            if plan is not None:
                self.retired.append(plan)
                del self.retired[:-self.MAX_RETIRED]
            plan = self.plans[key] = self._build(B, K, src.device, ptrs, wsqs)
        self._last = plan
        H.check(H.lib.vsp_style_plan_f32(plan["table"].data_ptr(), len(self.recorded), src.data_ptr(), B, src.stride(0), K, plan["max_cin"],
                                         plan["max_cout"], 1e-8, H._stream()), "style_plan")
        self.hits = plan["out"]

    def _build(self, B, K, device, ptrs, wsqs):
        n = sum(B * e[1].weight.shape[0] + (0 if q is None else B * q.shape[0]) for e, q in zip(self.recorded, wsqs))
        buf = torch.empty(n, device=device, dtype=torch.float32)
        arr = (_StyleLayerC * len(self.recorded))()
        out, off, max_cin, max_cout = {}, 0, 1, 0
        for i, ((owner, eql, src_off, _has, wscale), wsq) in enumerate(zip(self.recorded, wsqs)):
            cin = eql.weight.shape[0]
            mod = buf[off:off + B * cin].view(B, cin)
            off += B * cin
            demod = None
            if wsq is not None:
                cout = wsq.shape[0]
                demod = buf[off:off + B * cout].view(B, cout)
                off += B * cout
                max_cout = max(max_cout, cout)
            max_cin = max(max_cin, cin)
            a = arr[i]
            a.w, a.bias = eql.weight.data_ptr(), (eql.bias.data_ptr() if eql.bias is not None else None)
            a.wsq, a.mod, a.demod = (wsq.data_ptr() if wsq is not None else None), mod.data_ptr(), (demod.data_ptr() if demod is not None else None)
            a.src_off, a.cin, a.cout = src_off, cin, (wsq.shape[0] if wsq is not None else 0)
            w32 = ctypes.c_float(wscale if wsq is not None else 0.0).value     # (vsp_demod_f32 squares its float argument IN float: the same bits here)
            a.alpha, a.bias_scale, a.wscale2 = eql.scale, eql.lr_mul, ctypes.c_float(w32 * w32).value
            out[id(owner)] = (src_off, mod, demod, eql, 0 if wsq is None else wsq.data_ptr())
        table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        return {"ptrs": ptrs, "table": table, "buf": buf, "out": out, "max_cin": max_cin, "max_cout": max_cout, "wsqs": wsqs}

    def _row_offset(self, style):
        """element offset of the row view `style` = src[:, i] inside src, or None"""
        src = self.src
        if style.dim() != 2 or style.shape != (src.shape[0], src.shape[2]) or style.stride(1) != 1 or (style.shape[0] > 1 and style.stride(0) != src.stride(0)):
            return None
        d = style.data_ptr() - src.data_ptr()
        if d < 0 or d % (4 * src.shape[2]) or d // (4 * src.shape[2]) >= src.shape[1]:
            return None
        return d // 4

    def terms(self, owner, eql, style, wsq, wscale):
        off = self._row_offset(style)
        if self.hits is not None and off is not None:
            hit = self.hits.get(id(owner))
            # the plan evaluated THIS layer from THIS row with the tap sums the caller holds now -- anything else takes the two launches
            if hit is not None and hit[0] == off and hit[3] is eql and hit[4] == (0 if wsq is None else wsq.data_ptr()):
                return hit[1], hit[2]
        mod = eql(style)
        demod = H.demod_coefs(mod, wsq, wscale) if wsq is not None else None
        if self.rec is not None:
            ok = (off is not None and eql.activation is None and eql.weight.shape[1] == self.src.shape[2] and eql.weight.is_contiguous()
                  and eql.weight.data_ptr() % 16 == 0 and eql.weight.shape[0] <= 65535
                  and (wsq is None or (hasattr(owner, "_style_wsq") and owner._style_wsq() is wsq)))
            self.rec.append((owner, eql, off, wsq is not None, wscale) if ok else None)
        return mod, demod

    def end(self):
        if self.rec is not None and self.rec and all(e is not None for e in self.rec) and len({id(e[0]) for e in self.rec}) == len(self.rec):
            self.recorded = self.rec
        self.rec = self.hits = self.src = None


class style_context:
    """with style_context(module, "tag", latent): ... -- the modulation layers called inside take their vectors from the module's plan"""

    def __init__(self, module, tag, src):
        store = module.__dict__.setdefault("_style_ctx", {})
        self.ctx = store.get(tag)
        if self.ctx is None:
            self.ctx = store[tag] = StyleContext()
        self.src = src

    def __enter__(self):
        global _STYLE_CTX
        self.prev = _STYLE_CTX
        self.ctx.begin(self.src)
        _STYLE_CTX = self.ctx
        return self.ctx

    def __exit__(self, *exc):
        global _STYLE_CTX
        self.ctx.end()
        _STYLE_CTX = self.prev
        return False


def style_terms(owner, eql, style, wsq=None, wscale=None):
    """(modulation vector, demodulation coefficients or None) of one modulated layer: from the active plan, else two launches."""
    ctx = _STYLE_CTX
    if ctx is not None:
        return ctx.terms(owner, eql, style, wsq, wscale)
    mod = eql(style)
    return mod, (H.demod_coefs(mod, wsq, wscale) if wsq is not None else None)


class PixelNorm(nn.Module):
    def forward(self, x):
        return H.pixelnorm_dim1(x.contiguous())


class EqualLinear(_Cached):
    """weight stored divided by lr_mul; y = x (W * lr_mul/sqrt(in))^T + b * lr_mul, optional fused leaky-ReLU*sqrt2
    (reference models/RestoreNet.py:142-171)."""

    def __init__(self, in_dim, out_dim, bias=True, bias_init=0, lr_mul=1, activation=None):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim).fill_(bias_init)) if bias else None
        self.activation = activation
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def forward(self, x):
        return H.linear(x, self.weight, self.bias, alpha=self.scale, bias_scale=self.lr_mul,
                        act=1 if self.activation else 0)


class EqualConv2d(nn.Module):
    """Parameter holder (weight scaled by 1/sqrt(fan_in) at use); executed by its parent block."""

    def __init__(self, in_channel, out_channel, kernel_size, stride=1, padding=0, bias=True, dilation=1):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_channel, in_channel, kernel_size, kernel_size))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.bias = nn.Parameter(torch.zeros(out_channel)) if bias else None


class LeakyBias(nn.Module):
    """Holds the `bias` of a FusedLeakyReLU; the activation itself is fused into the producer's epilogue."""

    def __init__(self, channel):
        super().__init__()
        self.bias = nn.Parameter(torch.zeros(channel))


class NoiseInjection(nn.Module):
    def __init__(self):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(1))

    def draw(self, noise, like_shape, device):
        """explicit noise (B,1,H,W) or a fresh N(0,1) draw on the device (reference models/RestoreNet.py:564-569)."""
        if noise is None:
            return torch.randn(like_shape, device=device, dtype=torch.float32)
        return noise.contiguous()


class Blur(nn.Module):
    def __init__(self, kernel, pad, upsample_factor=1):
        super().__init__()
        kernel = make_kernel(kernel)
        if upsample_factor > 1:
            kernel = kernel * (upsample_factor ** 2)
        self.register_buffer("kernel", kernel)
        self.pad = pad


class Upsample(nn.Module):
    def __init__(self, kernel, factor=2):
        super().__init__()
        self.factor = factor
        kernel = make_kernel(kernel) * (factor ** 2)
        self.register_buffer("kernel", kernel)
        p = kernel.shape[0] - factor
        self.pad = ((p + 1) // 2 + factor - 1, p // 2)

    def forward(self, x):
        from .op import upfirdn2d
        return upfirdn2d(x.contiguous(), self.kernel, up=self.factor, down=1, pad=self.pad)


class ModulatedConv2d(_Cached):
    """weight (1, Cout, Cin, k, k) + `modulation` EqualLinear(style_dim -> Cin, bias_init=1) (+ `blur.kernel`)
    (reference models/RestoreNet.py:421-476, e4e/models/stylegan2/model.py:182-236)."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, demodulate=True, upsample=False, downsample=False,
                 blur_kernel=(1, 3, 3, 1), own_modulation=True, dilation=1):
        super().__init__()
        self.in_channel, self.out_channel, self.kernel_size = in_channel, out_channel, kernel_size
        self.upsample, self.downsample, self.demodulate, self.dilation = upsample, downsample, demodulate, dilation
        if upsample:
            self.blur = Blur(list(blur_kernel), pad=(1, 1), upsample_factor=2)
        if downsample:
            self.blur = Blur(list(blur_kernel), pad=(2, 2))
        self.scale = 1 / math.sqrt(in_channel * kernel_size ** 2)
        self.padding = ((kernel_size - 1) * dilation) // 2
        self.weight = nn.Parameter(torch.randn(1, out_channel, in_channel, kernel_size, kernel_size))
        if own_modulation:
            self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)

    # ---- derived tensors
    def _w_scaled(self):
        return (self.weight[0] * self.scale).contiguous()

    def packed(self):
        def build():
            w = H.pack_weight(self.weight[0], scale=self.scale)
            if self.upsample:  # one-pass stride-2 transposed conv: ordinary 3x3 packing
                return H.PackedConv(w, 1, self.out_channel, self.in_channel, 3, 3, 1, (1,), (1,))
            return H.PackedConv(w, 1, self.out_channel, self.in_channel, self.kernel_size, self.kernel_size,
                                2 if self.downsample else 1, (self.dilation,), (0 if self.downsample else self.padding,))
        return self._derive("packed", [self.weight], build)

    def wsq(self):
        return self._derive_static("wsq", [self.weight], lambda: (self.weight[0] ** 2).sum((2, 3)).contiguous())

    def _style_wsq(self):        # what layers.StyleContext bakes into its table (None: no demodulation)
        return self.wsq() if self.demodulate else None

    def demod(self, mod):
        return H.demod_coefs(mod, self.wsq(), self.scale) if self.demodulate else None

    def run(self, x, style, noise=None, noise_w=None, act_bias=None, ch_bias=None, res1=None, res2=None):
        """conv (+ the caller's fused tail).  `style` is the un-modulated style vector (B, style_dim)."""
        mod, demod = style_terms(self, self.modulation, style, self._style_wsq(), self.scale)
        x = x.contiguous()
        act = act_bias is not None
        if self.upsample:
            y = H.conv_transpose2d_s2_fused(x, self.packed(), in_scale=mod, out_scale=demod)
            return H.blur_fused(y, self.blur.kernel, self.blur.pad, noise=noise, noise_w=noise_w, act_bias=act_bias, act=act,
                                res1=res1, res2=res2)
        if self.downsample:
            x = H.blur_fused(x, self.blur.kernel, self.blur.pad)
        return H.conv2d_packed(x, self.packed(), in_scale=mod, out_scale=demod, ch_bias=ch_bias, noise=noise, noise_w=noise_w,
                               act2=1 if act else 0, bias2=act_bias, res1=res1, res2=res2)


class StyledConv(nn.Module):
    """conv -> noise -> bias + leaky-ReLU, all in one epilogue (reference models/RestoreNet.py:571-643)."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, upsample=False, downsample=False,
                 blur_kernel=(1, 3, 3, 1), demodulate=True):
        super().__init__()
        self.conv = ModulatedConv2d(in_channel, out_channel, kernel_size, style_dim, upsample=upsample, downsample=downsample,
                                    blur_kernel=blur_kernel, demodulate=demodulate)
        self.noise = NoiseInjection()
        self.activate = LeakyBias(out_channel)

    def out_hw(self, h, w):
        if self.conv.upsample:
            return 2 * h, 2 * w
        if self.conv.downsample:
            return h // 2, w // 2
        return h, w

    def forward(self, x, style, noise=None, res1=None, res2=None):
        oh, ow = self.out_hw(x.shape[2], x.shape[3])
        nz = self.noise.draw(noise, (x.shape[0], 1, oh, ow), x.device)
        return self.conv.run(x, style, noise=nz, noise_w=self.noise.weight, act_bias=self.activate.bias, res1=res1, res2=res2)


class ToRGB(nn.Module):
    """1x1 modulated conv without demodulation + bias + FIR-upsampled skip, skip add fused as the conv's residual
    (reference models/RestoreNet.py:647-666)."""

    def __init__(self, in_channel, style_dim, upsample=True, blur_kernel=(1, 3, 3, 1)):
        super().__init__()
        if upsample:
            self.upsample = Upsample(list(blur_kernel))
        self.conv = ModulatedConv2d(in_channel, 3, 1, style_dim, demodulate=False)
        self.bias = nn.Parameter(torch.zeros(1, 3, 1, 1))

    def forward(self, x, style, skip=None):
        """Cin -> 3 is an HBM stream, not a GEMM: the pointwise kernel reads every input plane once with 16-byte accesses."""
        conv = self.conv
        w = conv._derive("w_rgb", [conv.weight], lambda: (conv.weight[0, :, :, 0, 0] * conv.scale).contiguous())
        up = {}
        if skip is not None:  # Upsample(skip) (factor 2, 4x4 kernel, pad (2,1)) is evaluated inside the kernel
            u = self.upsample
            if u.factor == 2 and tuple(u.kernel.shape) == (4, 4) and u.pad == (2, 1) and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
                up = dict(up_src=skip.contiguous(), up_kernel=u.kernel)
            else:
                up = dict(res=u(skip))
        return H.pointwise(x.contiguous(), w, in_scale=style_terms(conv, conv.modulation, style)[0], ch_bias=self.bias.view(3), **up)


class SMARTLayer(_Cached):
    """SMART_layer (reference models/RestoreNet.py:179-244): shared modulation, four dilated modulated 3x3 branches
    (one launch, four dilation groups writing channel slices), `fusion` 3x3 conv whose epilogue carries
    bias+lrelu -> noise -> bias+lrelu."""

    def __init__(self, in_channel, out_channel, kernel_size, style_dim, blur_kernel=(1, 3, 3, 1)):
        super().__init__()
        self.in_channel, self.out_channel = in_channel, out_channel
        self.ModulatedConv2ds = nn.ModuleList(
            ModulatedConv2d(in_channel, out_channel // len(RATES), kernel_size, style_dim, own_modulation=False, dilation=r)
            for r in RATES)
        self.modulation = EqualLinear(style_dim, in_channel, bias_init=1)
        self.fusion = nn.Sequential(EqualConv2d(out_channel, out_channel, 3, padding=1, bias=False), LeakyBias(out_channel))
        self.noise = NoiseInjection()
        self.activate = LeakyBias(out_channel)

    def _branch_pack(self):
        ws = [m.weight for m in self.ModulatedConv2ds]

        def build():
            scale = self.ModulatedConv2ds[0].scale
            wp = H.pack_weight_stack([w[0] for w in ws], scale=scale)
            pc = H.PackedConv(wp, len(RATES), self.out_channel // len(RATES), self.in_channel, 3, 3, 1, RATES, RATES)
            return pc
        return self._derive("branches", ws, build)

    def _style_wsq(self):        # squared-tap sums of the four branches, (out_channel, in_channel); address stable (layers.StyleContext)
        ws = [m.weight for m in self.ModulatedConv2ds]
        return self._derive_static("branch_wsq", ws, lambda: torch.cat([(w[0] ** 2).sum((2, 3)) for w in ws], 0).contiguous())

    def _fusion_pack(self):
        conv = self.fusion[0]
        return self._derive("fusion", [conv.weight], lambda: H.PackedConv(
            H.pack_weight(conv.weight, scale=conv.scale), 1, self.out_channel, self.out_channel, 3, 3, 1, (1,), (1,)))

    def forward(self, x, style, noise=None):
        x = x.contiguous()
        pc = self._branch_pack()
        mod, demod = style_terms(self, self.modulation, style, self._style_wsq(), self.ModulatedConv2ds[0].scale)
        mid = H.conv2d_packed(x, pc, in_scale=mod, out_scale=demod)
        nz = self.noise.draw(noise, (x.shape[0], 1, x.shape[2], x.shape[3]), x.device)
        return H.conv2d_packed(mid, self._fusion_pack(), act1=True, bias1=self.fusion[1].bias, noise=nz,
                               noise_w=self.noise.weight, act2=1, bias2=self.activate.bias)


class LargeConvLayer(_Cached):
    """LargeConvLayer(downsample=False) (reference models/RestoreNet.py:725-787): four dilated EqualConv2d branches in one
    launch, 1x1 `fusion` conv with both FusedLeakyReLUs in its epilogue."""

    def __init__(self, in_channel, out_channel, kernel_size):
        super().__init__()
        self.in_channel, self.out_channel, self.kernel_size = in_channel, out_channel, kernel_size
        self.dilated_convs = nn.ModuleList(
            EqualConv2d(in_channel, out_channel // len(RATES), kernel_size, padding=((kernel_size - 1) * r) // 2, bias=False,
                        dilation=r) for r in RATES)
        self.fusion = nn.Sequential(EqualConv2d(out_channel, out_channel, 1, bias=False), LeakyBias(out_channel))
        self.activate = LeakyBias(out_channel)

    def _packs(self):
        ws = [m.weight for m in self.dilated_convs]

        def build():
            k = self.kernel_size
            wp = H.pack_weight_stack([m.weight for m in self.dilated_convs], scale=self.dilated_convs[0].scale)
            pads = tuple(m.padding for m in self.dilated_convs)
            pc = H.PackedConv(wp, len(RATES), self.out_channel // len(RATES), self.in_channel, k, k, 1, RATES, pads)
            f = self.fusion[0]
            pf = H.PackedConv(H.pack_weight(f.weight, scale=f.scale), 1, self.out_channel, self.out_channel, 1, 1, 1,
                              (1,), (0,))
            return pc, pf
        return self._derive("packs", ws + [self.fusion[0].weight], build)

    def _pointwise_weight(self):
        """kernel_size 1: branches and fusion are both 1x1 and linear, so the layer is ONE (out x in) matrix followed by the
        two FusedLeakyReLUs -- no (B, out, H, W) intermediate (the 3 -> 64 input layer at 512^2: 0.5 GB less traffic)."""
        ws = [m.weight for m in self.dilated_convs] + [self.fusion[0].weight]

        def build():
            wd = torch.cat([m.weight[:, :, 0, 0] * m.scale for m in self.dilated_convs], 0)   # (out, in)
            wf = self.fusion[0].weight[:, :, 0, 0] * self.fusion[0].scale                     # (out, out)
            return (wf.double() @ wd.double()).float().contiguous()
        return self._derive("pointwise", ws, build)

    def forward(self, x):
        if self.kernel_size == 1 and self.in_channel <= 4:
            return H.pointwise(x.contiguous(), self._pointwise_weight(), bias1=self.fusion[1].bias, bias2=self.activate.bias)
        pc, pf = self._packs()
        mid = H.conv2d_packed(x.contiguous(), pc)
        return H.conv2d_packed(mid, pf, act1=True, bias1=self.fusion[1].bias, act2=1, bias2=self.activate.bias)
