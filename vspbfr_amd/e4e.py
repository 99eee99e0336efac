"""The e4e/pSp side of the path on gfx950: style encoder (IR-SE50 FPN + 18 map2style heads), the StyleGAN2 prior
decoder with feature taps, and the `E4e_embedding` wrapper -- API and checkpoint layout of the reference's
Loss/e4e_embedding.py:71-135, e4e/models/psp.py:73-248, e4e/models/encoders/psp_encoders.py:124-200,
e4e/models/encoders/helpers.py:57-140 and e4e/models/stylegan2/model.py:366-552.

Execution notes (MI355X-first, not the reference's op sequence):
* eval-mode BatchNorm never runs as an op: the BN in front of a conv becomes the conv kernel's per-channel input
  scale/shift (applied to in-image samples only, so zero padding stays zero exactly as BN -> pad does in the reference),
  the BN behind a conv becomes the epilogue's per-channel scale/bias; PReLU / LeakyReLU(0.01) / conv bias ride in the
  same epilogue.
* the squeeze-excite gate and the residual add are one pass (x * gate + shortcut).
* bilinear 512->256 (align_corners=False, exact 2x2 mean) and AdaptiveAvgPool 1024->512 are the same 2x2 mean kernel.
"""
import math
from argparse import Namespace

import torch
from torch import nn

from . import hip_ops as H
from .layers import EqualLinear, NoiseInjection, PixelNorm, StyledConv, ToRGB, _Cached, style_context

IR50_BLOCKS = ((64, 64, 3), (64, 128, 4), (128, 256, 14), (256, 512, 3))


# ------------------------------------------------------------------------------------------------ StyleGAN2 prior
class ConstantInput(nn.Module):
    def __init__(self, channel, size=4):
        super().__init__()
        self.input = nn.Parameter(torch.randn(1, channel, size, size))

    def forward(self, batch):
        # a constant: the batch copy is built once per (parameter version, batch size), not once per call
        stamp = (self.input.data_ptr(), self.input._version, self.input.device, batch)
        hit = self.__dict__.get("_rep")
        if hit is None or hit[0] != stamp:
            hit = (stamp, self.input.detach().repeat(batch, 1, 1, 1))
            self.__dict__["_rep"] = hit
        return hit[1]


class Generator(nn.Module):
    """StyleGAN2 synthesis network used as the visual-style-prompt prior.  forward(styles, ..., return_features=True)
    returns (image, [feature after conv1 and after every up-conv]) like the reference (stylegan2/model.py:475-552)."""

    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], lr_mlp=0.01):
        super().__init__()
        self.size, self.style_dim = size, style_dim
        cm = channel_multiplier
        self.channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * cm, 128: 128 * cm, 256: 64 * cm, 512: 32 * cm,
                         1024: 16 * cm}
        ch, bk = self.channels, tuple(blur_kernel)
        self.style = nn.Sequential(PixelNorm(), *[EqualLinear(style_dim, style_dim, lr_mul=lr_mlp, activation="fused_lrelu")
                                                 for _ in range(n_mlp)])
        self.input = ConstantInput(ch[4])
        self.conv1 = StyledConv(ch[4], ch[4], 3, style_dim, blur_kernel=bk)
        self.to_rgb1 = ToRGB(ch[4], style_dim, upsample=False)
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.n_latent = self.log_size * 2 - 2
        self.convs, self.upsamples, self.to_rgbs, self.noises = nn.ModuleList(), nn.ModuleList(), nn.ModuleList(), nn.Module()
        for layer_idx in range(self.num_layers):
            res = (layer_idx + 5) // 2
            self.noises.register_buffer(f"noise_{layer_idx}", torch.randn(1, 1, 2 ** res, 2 ** res))
        in_ch = ch[4]
        for i in range(3, self.log_size + 1):
            out_ch = ch[2 ** i]
            self.convs.append(StyledConv(in_ch, out_ch, 3, style_dim, upsample=True, blur_kernel=bk))
            self.convs.append(StyledConv(out_ch, out_ch, 3, style_dim, blur_kernel=bk))
            self.to_rgbs.append(ToRGB(out_ch, style_dim))
            in_ch = out_ch

    def get_latent(self, z):
        return self.style(z)

    @torch.no_grad()
    def forward(self, styles, return_latents=False, inject_index=None, truncation=1, truncation_latent=None,
                input_is_latent=False, noise=None, randomize_noise=True, return_features=False, max_feature_res=None):
        """`max_feature_res` (extension): stop after the level of that resolution -- the restoration net only consumes
        features up to its own size, the 1024^2 tail only feeds the optional `style_sample` picture."""
        if not input_is_latent:
            styles = [self.style(s.contiguous()) for s in styles]
        if noise is None:
            noise = [None] * self.num_layers if randomize_noise else [getattr(self.noises, f"noise_{i}").expand(
                styles[0].shape[0], -1, -1, -1).contiguous() for i in range(self.num_layers)]
        if truncation < 1:
            styles = [truncation_latent + truncation * (s - truncation_latent) for s in styles]
        if len(styles) < 2:
            latent = styles[0].unsqueeze(1).repeat(1, self.n_latent, 1) if styles[0].ndim < 3 else styles[0]
        else:
            import random
            if inject_index is None:
                inject_index = random.randint(1, self.n_latent - 1)
            latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                                styles[1].unsqueeze(1).repeat(1, self.n_latent - inject_index, 1)], 1)
        B = latent.shape[0]
        with style_context(self, "gen", latent):   # (every layer's modulation / demodulation vector in two launches, layers.StyleContext)
            out = self.conv1(self.input(B), latent[:, 0], noise[0])
            skip = self.to_rgb1(out, latent[:, 1])
            feats = [out] if return_features else None
            i = 1
            for j in range(self.log_size - 2):
                if max_feature_res is not None and 2 ** (j + 3) > max_feature_res:
                    break
                out = self.convs[2 * j](out, latent[:, i], noise[1 + 2 * j])
                if return_features:
                    feats.append(out)
                out = self.convs[2 * j + 1](out, latent[:, i + 1], noise[2 + 2 * j])
                skip = self.to_rgbs[j](out, latent[:, i + 2], skip)
                i += 2
        if return_latents:
            return skip, latent
        return skip, feats


# ------------------------------------------------------------------------------------------------ IR-SE50 encoder
HANDOFF = None          # set by pipeline.run_batches around get_w_plus: an object with .point ("h": inside the encoder) and .go(tensors)
HEADS_BF16_ACT = H.tune_env("VSP_HEADS_BF16_ACT", "1") != "0"   # bf16 configuration: head-stage outputs as bf16 in HBM (Encoder4Editing._head_stage)
MAIN_STAGE_MIN = 16     # head stages whose OUTPUT map is at least this wide stay on the caller's stream (see Encoder4Editing.forward)


def _bn_fold(bn):
    a = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return a.contiguous(), (bn.bias - bn.running_mean * a).contiguous()


def _plain_pack(conv):
    w = conv.weight
    return H.PackedConv(H.pack_weight(w.contiguous()), 1, w.shape[0], w.shape[1], w.shape[2], w.shape[3], conv.stride[0],
                        (1,), (conv.padding[0],))


class SEModule(nn.Module):
    def __init__(self, channels, reduction):
        super().__init__()
        self.fc1 = nn.Conv2d(channels, channels // reduction, kernel_size=1, padding=0, bias=False)
        self.fc2 = nn.Conv2d(channels // reduction, channels, kernel_size=1, padding=0, bias=False)

    def gate(self, x):
        m = H.plane_mean(x)
        h = H.gemm_nt(m, self.fc1.weight.view(self.fc1.weight.shape[0], -1), act=1, slope=0.0, gain=1.0)  # ReLU
        return H.gemm_nt(h, self.fc2.weight.view(self.fc2.weight.shape[0], -1), act=2)                   # sigmoid


class bottleneck_IR_SE(_Cached):
    """res_layer = [BN, conv3x3, PReLU, conv3x3(stride), BN, SE]; shortcut = MaxPool2d(1, stride) or conv1x1+BN
    (reference helpers.py:89-113).  Indices of `res_layer` / `shortcut_layer` match the checkpoint keys."""

    def __init__(self, in_channel, depth, stride):
        super().__init__()
        self.stride, self.same = stride, in_channel == depth
        if self.same:
            self.shortcut_layer = nn.MaxPool2d(1, stride)
        else:
            self.shortcut_layer = nn.Sequential(nn.Conv2d(in_channel, depth, (1, 1), stride, bias=False), nn.BatchNorm2d(depth))
        self.res_layer = nn.Sequential(
            nn.BatchNorm2d(in_channel), nn.Conv2d(in_channel, depth, (3, 3), (1, 1), 1, bias=False), nn.PReLU(depth),
            nn.Conv2d(depth, depth, (3, 3), stride, 1, bias=False), nn.BatchNorm2d(depth), SEModule(depth, 16))

    def _consts(self):
        r = self.res_layer
        src = [r[0].weight, r[0].bias, r[0].running_mean, r[0].running_var, r[1].weight, r[3].weight, r[4].weight, r[4].bias,
               r[4].running_mean, r[4].running_var]
        if not self.same:
            s = self.shortcut_layer
            src += [s[0].weight, s[1].weight, s[1].bias, s[1].running_mean, s[1].running_var]

        def build():
            d = {"bn0": _bn_fold(r[0]), "c1": _plain_pack(r[1]), "c2": _plain_pack(r[3]), "bn4": _bn_fold(r[4])}
            if not self.same:
                d["sc"] = _plain_pack(self.shortcut_layer[0])
                d["scbn"] = _bn_fold(self.shortcut_layer[1])
            return d
        return self._derive("consts", src, build)

    def forward(self, x):
        c = self._consts()
        r = self.res_layer
        if self.same:
            shortcut = x if self.stride == 1 else H.subsample(x, self.stride)
        else:
            shortcut = H.conv2d_packed(x, c["sc"], ch_scale=c["scbn"][0], ch_bias=c["scbn"][1])
        y = H.conv2d_packed(x, c["c1"], in_scale=c["bn0"][0], in_scale_per_sample=False, in_shift=c["bn0"][1], act2=2,
                            prelu=r[2].weight)
        y = H.conv2d_packed(y, c["c2"], ch_scale=c["bn4"][0], ch_bias=c["bn4"][1])
        return H.scale_add(y, r[5].gate(y), shortcut)


class GradualStyleBlock(_Cached):
    """map2style head: log2(spatial) stride-2 3x3 convs with LeakyReLU(0.01), then EqualLinear (psp_encoders.py:34-55)."""

    def __init__(self, in_c, out_c, spatial):
        super().__init__()
        self.out_c, self.spatial = out_c, spatial
        mods = []
        c = in_c
        for _ in range(int(math.log2(spatial))):
            mods += [nn.Conv2d(c, out_c, kernel_size=3, stride=2, padding=1), nn.LeakyReLU()]
            c = out_c
        self.convs = nn.Sequential(*mods)
        self.linear = EqualLinear(out_c, out_c, lr_mul=1)

    def forward(self, x):
        convs = [m for m in self.convs if isinstance(m, nn.Conv2d)]
        packs = self._derive("packs", [m.weight for m in convs], lambda: [_plain_pack(m) for m in convs])
        for m, pc in zip(convs, packs):
            x = H.conv2d_packed(x, pc, ch_bias=m.bias, act2=1, slope2=0.01, gain2=1.0)
        return self.linear(x.view(-1, self.out_c))


class Encoder4Editing(_Cached):
    def __init__(self, num_layers, mode="ir", opts=None):
        super().__init__()
        assert num_layers == 50 and mode == "ir_se", "the restoration path uses the IR-SE50 backbone only"
        in_ch = getattr(opts, "input_channel", 3)
        self.input_layer = nn.Sequential(nn.Conv2d(in_ch, 64, (3, 3), 1, 1, bias=False), nn.BatchNorm2d(64), nn.PReLU(64))
        units = []
        for in_c, depth, n in IR50_BLOCKS:
            units.append(bottleneck_IR_SE(in_c, depth, 2))
            units += [bottleneck_IR_SE(depth, depth, 1) for _ in range(n - 1)]
        self.body = nn.Sequential(*units)
        log_size = int(math.log(opts.stylegan_size, 2))
        self.style_count = 2 * log_size - 2
        self.coarse_ind, self.middle_ind = 3, 7
        self.styles = nn.ModuleList(
            GradualStyleBlock(512, 512, 16 if i < self.coarse_ind else 32 if i < self.middle_ind else 64)
            for i in range(self.style_count))
        self.latlayer1 = nn.Conv2d(256, 512, kernel_size=1, stride=1, padding=0)
        self.latlayer2 = nn.Conv2d(128, 512, kernel_size=1, stride=1, padding=0)

    @torch.no_grad()
    def forward(self, x, latent_avg=None):
        """`latent_avg` (extension): added to the codes in the same launch that assembles them (My_pSp.forward does it)."""
        il = self.input_layer
        src = [il[0].weight, il[1].weight, il[1].bias, il[1].running_mean, il[1].running_var, self.latlayer1.weight,
               self.latlayer2.weight]
        c = self._derive("consts", src, lambda: {"in": _plain_pack(il[0]), "bn": _bn_fold(il[1]),
                                                 "l1": _plain_pack(self.latlayer1), "l2": _plain_pack(self.latlayer2)})
        x = H.conv2d_packed(x.contiguous(), c["in"], ch_scale=c["bn"][0], ch_bias=c["bn"][1], act2=2, prelu=il[2].weight)
        taps = {}
        for i, unit in enumerate(self.body):
            x = unit(x)
            if i in (6, 20, 23):
                taps[i] = x
        c1, c2, c3 = taps[6], taps[20], taps[23]
        p2 = H.upsample_add(c3, H.conv2d_packed(c2, c["l1"], ch_bias=self.latlayer1.bias))
        p1 = H.upsample_add(p2, H.conv2d_packed(c1, c["l2"], ch_bias=self.latlayer2.bias))
        B = x.shape[0]
        # The map2style heads, class by class (coarse on c3, middle on p2, fine on p1).  Their first stages are chip-filling convolutions
        # (512 -> 5632 at 64^2 -> 32^2: 425 GFLOP at batch 8), the later ones run on maps of 8^2 and below in launches of a few dozen
        # workgroups.  With a HANDOFF (pipeline.run_batches) everything up to here and every stage whose output map is at least
        # MAIN_STAGE_MIN^2 stays on the caller's stream; the small-map stages, the final GEMMs and the code assembly go to the side stream,
        # where they (and the sampler chain behind them) run underneath the previous batch's big convolutions.
        runs = []
        for (lo, hi), f in zip(self._head_classes(), (c3, p2, p1)):
            stages, lw, lb = self._head_consts(lo, hi)
            runs.append([f, 0, stages, lw, lb, lo, hi])
        ho = HANDOFF
        if ho is not None and ho.point == "h":
            for r in runs:
                while r[1] < len(r[2]) and (r[0].shape[-1] // 2) >= MAIN_STAGE_MIN:
                    r[0] = self._head_stage(r[0], r[2][r[1]])
                    r[1] += 1
            ho.go([r[0] for r in runs])
        heads = torch.empty((self.style_count, B, 512), device=x.device, dtype=torch.float32)   # [w0, delta_1, ..., delta_17]
        lin = self.styles[0].linear
        for f, k, stages, lw, lb, lo, hi in runs:
            for st in stages[k:]:
                f = self._head_stage(f, st)
            f = H.to_f32(f)
            nh = hi - lo
            H.gemm_nt(f, lw, out=heads[lo:hi], dims=(nh, B, 512, 512), a_strides=(512, nh * 512, 1), b_strides=(512 * 512, 512, 1),
                      alpha=lin.scale, bias=lb, bias_scale=lin.lr_mul, bias_zs=512)
        return H.e4e_codes(heads, latent_avg)                  # w[:, i] = w0 + delta_i (psp_encoders.py:188-199), (B, 18, 512)

    def _head_classes(self):
        return ((0, self.coarse_ind), (self.coarse_ind, self.middle_ind), (self.middle_ind, self.style_count))

    @staticmethod
    def _head_stage(x, stage):
        """One stage of a head class.  In the bf16-kernel configuration (hip_ops.BF16_CONV is True) the stage outputs travel as bf16 -- the next
        stage's kernel rounds its input to bf16 while staging anyway, so the values are the same; the 512 -> 5632 stem then writes half the bytes
        and the grouped second stage stages pixel pairs (round 6; small maps fall back to the fp32 kernels, which convert)."""
        pc, bias = stage
        if HEADS_BF16_ACT and H.BF16_CONV is True and x.shape[-1] >= 32:
            prev = H.ACT_BF16
            H.ACT_BF16 = True
            try:
                return H.conv2d_packed(x, pc, ch_bias=bias, act2=1, slope2=0.01, gain2=1.0)
            finally:
                H.ACT_BF16 = prev
        return H.conv2d_packed(x, pc, ch_bias=bias, act2=1, slope2=0.01, gain2=1.0)

    def _head_consts(self, lo, hi):
        """All map2style heads fed by one feature map (same depth) as ONE launch per stage: the first conv is a plain conv
        with the heads' output channels concatenated, the following ones are true grouped convs (one group per head), the
        final EqualLinear a batched GEMM.  Returns (stages [(packed conv, bias)], stacked linear weights, biases)."""
        heads = [self.styles[i] for i in range(lo, hi)]
        nh = len(heads)
        convs = [[m for m in h.convs if isinstance(m, nn.Conv2d)] for h in heads]
        n_stage = len(convs[0])
        src = [m.weight for hc in convs for m in hc] + [m.bias for hc in convs for m in hc] + \
              [h.linear.weight for h in heads] + [h.linear.bias for h in heads]

        def build():
            stages = []
            for s_ in range(n_stage):
                ws = [hc[s_].weight for hc in convs]
                bias = torch.cat([hc[s_].bias for hc in convs]).contiguous()
                if s_ == 0:
                    wcat = torch.cat(ws, 0).contiguous()
                    pc = H.PackedConv(H.pack_weight(wcat), 1, nh * 512, wcat.shape[1], 3, 3, 2, (1,), (1,))
                else:
                    wp = torch.stack([H.pack_weight(w_.contiguous())[0] for w_ in ws]).contiguous()
                    pc = H.PackedConv(wp, nh, 512, 512, 3, 3, 2, (1,), (1,), x_group_stride=512 if nh > 1 else 0)
                stages.append((pc, bias))
            lw = torch.stack([h.linear.weight for h in heads]).contiguous()
            lb = torch.stack([h.linear.bias for h in heads]).contiguous()
            return stages, lw, lb
        return self._derive(f"heads{lo}", src, build)

    def _heads(self, lo, hi, feat, out=None):
        """One class of heads end to end: (hi - lo, B, 512)."""
        stages, lw, lb = self._head_consts(lo, hi)
        x = feat
        for st in stages:
            x = self._head_stage(x, st)
        x = H.to_f32(x)
        B, nh, lin = x.shape[0], hi - lo, self.styles[lo].linear
        return H.gemm_nt(x, lw, out=out, dims=(nh, B, 512, 512), a_strides=(512, nh * 512, 1), b_strides=(512 * 512, 512, 1),
                         alpha=lin.scale, bias=lb, bias_scale=lin.lr_mul, bias_zs=512)


# ------------------------------------------------------------------------------------------------ pSp wrapper
def get_keys(d, name):
    if "state_dict" in d:
        d = d["state_dict"]
    return {k[len(name) + 1:]: v for k, v in d.items() if k[:len(name)] == name}


class My_pSp(nn.Module):
    """reference e4e/models/psp.py:73-248 (the parts the restoration path touches)."""

    def __init__(self, opts, out_size, size, device, input_channel=3, use_generator=False, ckpt=None):
        super().__init__()
        self.opts = opts
        self.opts.input_channel = input_channel
        self.device = device
        if opts.encoder_type != "Encoder4Editing":
            raise Exception(f"{opts.encoder_type} is not a valid encoder for the restoration path")
        self.encoder = Encoder4Editing(50, "ir_se", self.opts)
        self.use_generator = use_generator
        self.decoder = Generator(opts.stylegan_size, 512, 8, channel_multiplier=2)
        self.latent_avg = None
        self.log_size = int(math.log(size, 2))
        self.n_latent = self.log_size * 2 - 2
        self.out_size = out_size
        self.out_n_latent = int(math.log(out_size, 2)) * 2 - 2
        if ckpt is None and getattr(opts, "checkpoint_path", None) is not None:
            ckpt = torch.load(opts.checkpoint_path, map_location="cpu")
        if ckpt is not None:
            self.load_weights(ckpt)
        self.style = self.decoder.style
        for p_ in self.parameters():
            p_.requires_grad = False

    def load_weights(self, ckpt):
        self.encoder.load_state_dict(get_keys(ckpt, "encoder"), strict=True)
        self.decoder.load_state_dict(get_keys(ckpt, "decoder"), strict=True)
        self.latent_avg = ckpt["latent_avg"].to(self.device) if "latent_avg" in ckpt else None

    @torch.no_grad()
    def forward(self, x):
        avg = self.latent_avg if self.opts.start_from_latent_avg else None
        if avg is not None and (avg.device != x.device or not avg.is_contiguous()):
            avg = self.latent_avg = avg.to(x.device).contiguous()
        codes = self.encoder(x, latent_avg=avg)
        return codes if codes.shape[1] == self.n_latent else codes[:, :self.n_latent]

    @torch.no_grad()
    def stylegan2_feat_forward(self, codes, resize=True, randomize_noise=True, return_features=True, noise=None,
                               with_sample=True):
        """`noise=` and `with_sample=` are extensions: explicit per-layer noise, and skipping the levels above out_size
        whose only consumer is the `style_sample` picture (then images is None)."""
        images, feats = self.decoder([codes], input_is_latent=True, randomize_noise=randomize_noise, noise=noise,
                                     return_features=return_features,
                                     max_feature_res=None if with_sample else self.out_size)
        feats = feats[:self.out_n_latent]
        if not with_sample:
            return None, feats
        if resize:
            while images.shape[-1] > self.out_size:  # AdaptiveAvgPool2d((out, out)) from a power-of-two multiple
                images = H.avgpool2x2(images)
        return images, feats


class E4e_embedding(nn.Module):
    """reference Loss/e4e_embedding.py:71-135.  `model_path` may also be an already-loaded checkpoint dict
    {'state_dict', 'latent_avg', 'opts'} (extension, used with synthetic weights)."""

    def __init__(self, model_path, out_size, size, device, input_channel=3, use_generator=False):
        super().__init__()
        ckpt = model_path if isinstance(model_path, dict) else torch.load(model_path, map_location="cpu")
        opts = dict(ckpt["opts"])
        opts["checkpoint_path"] = None if isinstance(model_path, dict) else model_path
        self.E4Enet = My_pSp(Namespace(**opts), out_size, size, device, input_channel=input_channel,
                             use_generator=use_generator, ckpt=ckpt).eval().to(device)

    @torch.no_grad()
    def get_w_plus(self, img, weight_map=None):
        # F.interpolate(img, (256, 256), mode="bilinear") of the reference: from 512^2 it is exactly the 2x2 mean
        if img.shape[-1] == 512 and img.shape[-2] == 512:
            return self.E4Enet(H.avgpool2x2(img.contiguous()))
        if img.shape[-1] == 256 and img.shape[-2] == 256:
            return self.E4Enet(img.contiguous())
        return self.E4Enet(H.resize_bilinear(img.contiguous(), (256, 256)))

    @torch.no_grad()
    def get_stylegan_feats(self, styles, noise=None, with_sample=True):
        return self.E4Enet.stylegan2_feat_forward(styles.contiguous(), resize=True, randomize_noise=True, noise=noise,
                                                  with_sample=with_sample)

    def get_stylegan_featsV2(self, styles, grad=False, return_feat=True, noise=None):
        """reference Loss/e4e_embedding.py:139-155 -> psp.stylegan2_feat_forward_v2 (:265-281): the prior's picture (pooled to
        out_size) from ALL the codes; grad=True keeps the graph to `styles` (code_diffuser_train.py:175 trains the Code_diffuser
        through the frozen decoder) and runs the differentiable forward of vspbfr_amd/training.py."""
        net = self.E4Enet
        if not grad:
            with torch.no_grad():
                images, feats = net.decoder([styles.contiguous()], input_is_latent=True, randomize_noise=True, noise=noise,
                                            return_features=True)
                while images.shape[-1] > net.out_size:
                    images = H.avgpool2x2(images)
            return (images, feats[:net.out_n_latent]) if return_feat else images
        if return_feat:
            raise RuntimeError("get_stylegan_featsV2(grad=True): only the picture is differentiable here (return_feat=False), as "
                               "code_diffuser_train.py:175 asks for it")
        from . import training
        return training.face_pool(training.generator_forward(net.decoder, styles[:, :net.decoder.n_latent], noise), net.out_size)

    def open_stylegan_grad(self):      # (the reference flips the decoder's requires_grad around the step without ever stepping it:
        pass                           #  the gradient to the codes does not need it, so these are no-ops here)

    def close_stylegan_grad(self):
        pass
