"""Restoration_net on gfx950: constructor/forward API and checkpoint layout of the reference's models/RestoreNet.py
(`Restoration_net(size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1,3,3,1], lr_mlp=0.01)`, forward signature
models/RestoreNet.py:968-981, state-dict keys SURVEY.md section 8a row 10), executed by the fused HIP blocks of
vspbfr_amd/layers.py.

Extension over the reference API: `enc_noise=` / `dec_noise=` lists give every NoiseInjection an explicit tensor (parity
runs); the reference's single `noise=` list cannot work (it is shared, reversed, between encoder and decoder whose
shapes differ: SURVEY.md section 8c), so passing `noise=` or `randomize_noise=False` raises instead of crashing later.
"""
import math
import random

import torch
from torch import nn

from . import hip_ops as H
from .layers import EqualLinear, LargeConvLayer, PixelNorm, SMARTLayer, StyledConv, ToRGB, style_context


def make_noise(batch, latent_dim, n_noise, device):
    if n_noise == 1:
        return torch.randn(batch, latent_dim, device=device)
    return torch.randn(n_noise, batch, latent_dim, device=device).unbind(0)


def mixing_noise(batch, latent_dim, prob, device):
    """reference restoration_test.py:77-82 / models/RestoreNet.py:17-22."""
    if prob > 0 and random.random() < prob:
        return make_noise(batch, latent_dim, 2, device)
    return [make_noise(batch, latent_dim, 1, device)]


class Restoration_net(nn.Module):
    def __init__(self, size, style_dim, n_mlp, channel_multiplier=2, blur_kernel=[1, 3, 3, 1], lr_mlp=0.01):
        super().__init__()
        self.size, self.style_dim = size, style_dim
        cm = channel_multiplier
        self.channels = {4: 512, 8: 512, 16: 512, 32: 512, 64: 256 * cm, 128: 128 * cm, 256: 64 * cm, 512: 32 * cm,
                         1024: 16 * cm}
        ch = self.channels
        bk = tuple(blur_kernel)
        self.log_size = int(math.log(size, 2))
        self.num_layers = (self.log_size - 2) * 2 + 1
        self.n_latent = self.log_size * 2 - 2

        # decoder (style = [W+ code, mapped noise code, x_global] = 4 * style_dim wide)
        self.conv1 = SMARTLayer(ch[4], ch[4], 3, 4 * style_dim, blur_kernel=bk)
        self.to_rgb1 = ToRGB(ch[4], 4 * style_dim, upsample=False)
        self.convs = nn.ModuleList()
        self.upsamples = nn.ModuleList()
        self.to_rgbs = nn.ModuleList()
        self.noises = nn.Module()
        self.style = nn.Sequential(PixelNorm(), *[EqualLinear(style_dim, style_dim, lr_mul=lr_mlp, activation="fused_lrelu")
                                                 for _ in range(n_mlp)])
        for layer_idx in range(self.num_layers):
            res = (layer_idx + 5) // 2
            self.noises.register_buffer(f"noise_{layer_idx}", torch.randn(1, 1, 2 ** res, 2 ** res))
        in_ch = ch[4]
        for i in range(3, self.log_size + 1):
            out_ch = ch[2 ** i]
            self.convs.append(StyledConv(in_ch, out_ch, 3, 4 * style_dim, upsample=True, blur_kernel=bk))
            self.convs.append(SMARTLayer(out_ch, out_ch, 3, 4 * style_dim, blur_kernel=bk))
            self.to_rgbs.append(ToRGB(out_ch, 4 * style_dim))
            in_ch = out_ch

        # encoder (style = [W+ code, mapped noise code] = 2 * style_dim wide)
        self.down_from_big = LargeConvLayer(3, ch[size], kernel_size=1)
        self.encoder_convs = nn.ModuleList()
        in_ch = ch[size]
        for i in range(self.log_size, 2, -1):
            mid, out_ch = ch[2 ** i], ch[2 ** (i - 1)]
            self.encoder_convs.append(SMARTLayer(in_ch, mid, 3, 2 * style_dim, blur_kernel=bk))
            self.encoder_convs.append(StyledConv(mid, out_ch, 3, 2 * style_dim, downsample=True, blur_kernel=bk))
            in_ch = out_ch
        self.final_layer = LargeConvLayer(in_ch, ch[4], kernel_size=3)
        self.final_linear = nn.Sequential(EqualLinear(ch[4] * 16, ch[4] * 2, activation="fused_lrelu"), nn.Dropout2d(0.5))
        self.final_transfer = EqualLinear(ch[4] * 2, ch[4] * 16, activation="fused_lrelu")

    # ------------------------------------------------------------------------------------------------------------
    def get_latent(self, z):
        return self.style(z)

    def mean_latent(self, n_latent, device):
        return self.style(torch.randn(n_latent, self.style_dim, device=device)).mean(0, keepdim=True)

    def _noise_latent(self, noise_styles, inject_index):
        if len(noise_styles) < 2:
            s = noise_styles[0]
            return s.unsqueeze(1).repeat(1, self.n_latent, 1) if s.ndim < 3 else s
        if inject_index is None:
            inject_index = random.randint(1, self.n_latent - 1)
        return torch.cat([noise_styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                          noise_styles[1].unsqueeze(1).repeat(1, self.n_latent - inject_index, 1)], 1)

    def encoder_forward(self, imgs, latent_cp, enc_noise):
        """U-shaped style-modulated encoder (reference models/RestoreNet.py:915-942); returns x_global and the features
        coarse -> fine."""
        B = imgs.shape[0]
        out = self.down_from_big(imgs)
        feats = []
        with style_context(self, "enc", latent_cp):   # every modulation / demodulation vector of the encoder in two launches (layers.StyleContext)
            for ii in range(0, len(self.encoder_convs), 2):
                sty = latent_cp[:, ii]
                out = self.encoder_convs[ii](out, sty, enc_noise[ii])
                feats.append(out)
                out = self.encoder_convs[ii + 1](out, sty, enc_noise[ii + 1])  # same latent index as the SMART layer
        out = self.final_layer(out)
        x_global = self.final_linear[0](out.view(B, -1))
        if self.training:   # Dropout2d(0.5) of final_linear (models/RestoreNet.py:907-909): the training loop samples fakes in train mode
            x_global = self.final_linear[1](x_global)
        early = self.final_transfer(x_global).view(B, -1, 4, 4)
        feats.append(H.add3(out, early))
        return x_global, feats[::-1]

    @torch.no_grad()
    def forward(self, images, de_feats, pre_styles, noise_styles, return_latents=False, inject_index=None, truncation=1,
                truncation_latent=None, input_is_latent=False, noise=None, randomize_noise=True, enc_noise=None,
                dec_noise=None):
        if noise is not None or not randomize_noise:
            raise RuntimeError("Restoration_net: the reference's shared `noise=` list / randomize_noise=False cannot be "
                               "served (encoder and decoder need different shapes, models/RestoreNet.py:1018); pass "
                               "enc_noise= and dec_noise= instead")
        # (always a no-grad forward: in train() mode it is the sampling pass of the training loop -- Dropout2d of final_linear active --
        #  the differentiable forward over these parameters is vspbfr_amd.training.restoration_net_forward)
        images = images.contiguous()
        if not input_is_latent:
            noise_styles = [self.style(s.contiguous()) for s in noise_styles]
        if truncation < 1:
            noise_styles = [truncation_latent + truncation * (s - truncation_latent) for s in noise_styles]
        B, nl, sd = images.shape[0], self.n_latent, self.style_dim
        pre_styles = pre_styles.contiguous()
        if len(noise_styles) == 1 and noise_styles[0].ndim == 2 and pre_styles.shape[2] == sd and pre_styles.shape[1] >= nl:
            # latent = [W+ code | mapped z], and its token-flipped copy for the encoder: two launches instead of repeat + cat + flip
            z = noise_styles[0].contiguous()
            latent = H.rows_concat(B, nl, [(pre_styles, sd, True, False), (z, sd, False, False)])
            latent_cp = H.rows_concat(B, nl, [(pre_styles, sd, True, True), (z, sd, False, False)])   # token t <- token nl - 1 - t
        else:
            noise_latent = self._noise_latent(noise_styles, inject_index)
            latent = torch.cat([pre_styles[:, :noise_latent.shape[1], :], noise_latent], dim=-1)
            latent_cp = torch.flip(latent, dims=[1])
        n_enc = len(self.encoder_convs)
        enc_noise = list(enc_noise) if enc_noise is not None else [None] * n_enc
        dec_noise = list(dec_noise) if dec_noise is not None else [None] * self.num_layers
        if len(enc_noise) != n_enc or len(dec_noise) != self.num_layers:
            raise RuntimeError(f"expected {n_enc} encoder and {self.num_layers} decoder noise tensors")

        x_global, feats = self.encoder_forward(images, latent_cp, enc_noise)

        # decoder styles cat[latent_i, x_global] for all layers in ONE concatenation; sty(i) is then a strided row view
        sty_all = H.rows_concat(latent.shape[0], latent.shape[1], [(latent, latent.shape[2], True, False),
                                                                    (x_global.contiguous(), x_global.shape[1], False, False)])

        def sty(i):
            return sty_all[:, i]

        with style_context(self, "dec", sty_all):   # (the decoder's modulation / demodulation vectors: two launches, layers.StyleContext)
            out = self.conv1(feats[0], sty(0), dec_noise[0])
            skip = self.to_rgb1(out, sty(1))
            i = 1
            for j in range(self.log_size - 2):
                k = (i + 1) // 2
                # up-conv; `out + feat + sty_de_feat` (models/RestoreNet.py:1035) rides in the blur epilogue
                out = self.convs[2 * j](out, sty(i), dec_noise[1 + 2 * j], res1=feats[k], res2=de_feats[k].contiguous())
                out = self.convs[2 * j + 1](out, sty(i + 1), dec_noise[2 + 2 * j])
                skip = self.to_rgbs[j](out, sty(i + 2), skip)
                i += 2
        if return_latents:
            return skip, latent
        return skip
