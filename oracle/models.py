"""Functional CPU restatement of the four networks of the restoration path (TEST INFRASTRUCTURE).

Every function takes a flat state dict `sd` (checkpoint key -> fp32 CPU tensor, the reference's own key layout,
SURVEY.md section 8b) and a key prefix `p`.  All random draws are explicit arguments (the reference draws them from
the global RNG: SURVEY.md section 8c lists the order).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .ops import equal_linear, fused_leaky_relu, modulated_conv, pixel_norm, upfirdn2d

RATES = (1, 2, 4, 8)


# --------------------------------------------------------------------------------------------- StyleGAN2-style blocks
def style_mlp(sd, p, z, n_mlp=8, lr_mlp=0.01):
    """PixelNorm + n_mlp x EqualLinear(lr_mul, fused_lrelu) -- reference models/RestoreNet.py:837-846 (keys style.1..8)."""
    x = pixel_norm(z)
    for i in range(1, n_mlp + 1):
        x = equal_linear(x, sd[f"{p}{i}.weight"], sd[f"{p}{i}.bias"], lr_mul=lr_mlp, activation=True)
    return x


def styled_conv(sd, p, x, style, noise, mode="same"):
    """StyledConv / StyledConv_down: modulated conv -> noise -> FusedLeakyReLU
    (reference models/RestoreNet.py:599-605,637-643; e4e/models/stylegan2/model.py:337-342)."""
    mod = equal_linear(style, sd[p + "conv.modulation.weight"], sd[p + "conv.modulation.bias"])
    blur = sd.get(p + "conv.blur.kernel")
    out = modulated_conv(x, sd[p + "conv.weight"], mod, mode=mode, blur_kernel=blur)
    out = out + sd[p + "noise.weight"] * noise
    return fused_leaky_relu(out, sd[p + "activate.bias"])


def to_rgb(sd, p, x, style, skip=None):
    """ToRGB: 1x1 modulated conv without demodulation + bias (+ 2x FIR-upsampled skip)
    (reference models/RestoreNet.py:657-666; e4e stylegan2/model.py:355-364; Upsample pad (2,1): RestoreNet.py:51-56)."""
    mod = equal_linear(style, sd[p + "conv.modulation.weight"], sd[p + "conv.modulation.bias"])
    out = modulated_conv(x, sd[p + "conv.weight"], mod, demodulate=False) + sd[p + "bias"]
    if skip is not None:
        out = out + upfirdn2d(skip, sd[p + "upsample.kernel"], up=2, down=1, pad=(2, 1))
    return out


def smart_layer(sd, p, x, style, noise):
    """SMART_layer.forward, reference models/RestoreNet.py:225-244: one shared modulation, four dilated modulated
    3x3 branches (Cout/4 each), concat, plain 3x3 `fusion` ConvLayer (+FusedLeakyReLU), noise, FusedLeakyReLU."""
    mod = equal_linear(style, sd[p + "modulation.weight"], sd[p + "modulation.bias"])
    outs = [modulated_conv(x, sd[f"{p}ModulatedConv2ds.{i}.weight"], mod, dilation=r) for i, r in enumerate(RATES)]
    out = torch.cat(outs, dim=1)
    w = sd[p + "fusion.0.weight"]
    out = F.conv2d(out, w * (1 / math.sqrt(w.shape[1] * w.shape[2] ** 2)), padding=w.shape[2] // 2)
    out = fused_leaky_relu(out, sd[p + "fusion.1.bias"])
    out = out + sd[p + "noise.weight"] * noise
    return fused_leaky_relu(out, sd[p + "activate.bias"])


def large_conv_layer(sd, p, x):
    """LargeConvLayer.forward (downsample=False), reference models/RestoreNet.py:773-787: four dilated EqualConv2d
    (no bias), concat, 1x1 `fusion` ConvLayer with FusedLeakyReLU, then a second FusedLeakyReLU."""
    outs = []
    for i, r in enumerate(RATES):
        w = sd[f"{p}dilated_convs.{i}.weight"]
        k = w.shape[2]
        outs.append(F.conv2d(x, w * (1 / math.sqrt(w.shape[1] * k * k)), padding=((k - 1) * r) // 2, dilation=r))
    out = torch.cat(outs, dim=1)
    w = sd[p + "fusion.0.weight"]
    out = F.conv2d(out, w * (1 / math.sqrt(w.shape[1])), padding=0)
    out = fused_leaky_relu(out, sd[p + "fusion.1.bias"])
    return fused_leaky_relu(out, sd[p + "activate.bias"])


# --------------------------------------------------------------------------------------------- Restoration_net
def restoration_noise_shapes(size, batch):
    """Shapes of the NoiseInjection draws in call order (reference models/RestoreNet.py:564-569 via :922-927 and
    :1022-1037): encoder = SMART at r then down-conv at r/2 for r = size..8; decoder = 4, then (r, r) for r = 8..size."""
    log_size = int(math.log2(size))
    enc = []
    for i in range(log_size, 2, -1):
        enc += [(batch, 1, 2 ** i, 2 ** i), (batch, 1, 2 ** (i - 1), 2 ** (i - 1))]
    dec = [(batch, 1, 4, 4)]
    for i in range(3, log_size + 1):
        dec += [(batch, 1, 2 ** i, 2 ** i)] * 2
    return enc, dec


def restoration_net(sd, size, images, de_feats, pre_styles, noise_styles, enc_noise, dec_noise, inject_index=None):
    """Restoration_net.forward, reference models/RestoreNet.py:968-1046 (+ encoder_forward :915-942)."""
    B = images.shape[0]
    log_size = int(math.log2(size))
    n_latent = log_size * 2 - 2
    styles = [style_mlp(sd, "style.", z) for z in noise_styles]
    if len(styles) < 2:
        noise_latent = styles[0].unsqueeze(1).repeat(1, n_latent, 1)
    else:
        assert inject_index is not None, "two noise codes need an explicit inject_index (reference draws random.randint)"
        noise_latent = torch.cat([styles[0].unsqueeze(1).repeat(1, inject_index, 1),
                                  styles[1].unsqueeze(1).repeat(1, n_latent - inject_index, 1)], 1)
    latent = torch.cat([pre_styles[:, :n_latent], noise_latent], dim=-1)
    latent_cp = torch.flip(latent, dims=[1])

    out = large_conv_layer(sd, "down_from_big.", images)
    feats = []
    for ii in range(0, 2 * (log_size - 2), 2):
        out = smart_layer(sd, f"encoder_convs.{ii}.", out, latent_cp[:, ii], enc_noise[ii])
        feats.append(out)
        out = styled_conv(sd, f"encoder_convs.{ii + 1}.", out, latent_cp[:, ii], enc_noise[ii + 1], mode="down")
    out = large_conv_layer(sd, "final_layer.", out)
    feats.append(out)
    x_global = equal_linear(out.reshape(B, -1), sd["final_linear.0.weight"], sd["final_linear.0.bias"], activation=True)
    early = equal_linear(x_global, sd["final_transfer.weight"], sd["final_transfer.bias"], activation=True)
    feats[-1] = feats[-1] + early.view(B, -1, 4, 4)
    feats = feats[::-1]

    def sty(i):
        return torch.cat([latent[:, i], x_global], dim=1)

    out = smart_layer(sd, "conv1.", feats[0], sty(0), dec_noise[0])
    skip = to_rgb(sd, "to_rgb1.", out, sty(1))
    i = 1
    for j in range(log_size - 2):
        out = styled_conv(sd, f"convs.{2 * j}.", out, sty(i), dec_noise[1 + 2 * j], mode="up")
        k = (i + 1) // 2
        out = out + feats[k] + de_feats[k]
        out = smart_layer(sd, f"convs.{2 * j + 1}.", out, sty(i + 1), dec_noise[2 + 2 * j])
        skip = to_rgb(sd, f"to_rgbs.{j}.", out, sty(i + 2), skip)
        i += 2
    return skip


# --------------------------------------------------------------------------------------------- e4e StyleGAN2 prior
def generator_noise_shapes(size, batch):
    """17 draws for size 1024: 4, then (r, r) for r = 8..size (reference e4e/models/stylegan2/model.py:287-292,526-540)."""
    log_size = int(math.log2(size))
    shapes = [(batch, 1, 4, 4)]
    for i in range(3, log_size + 1):
        shapes += [(batch, 1, 2 ** i, 2 ** i)] * 2
    return shapes


def stylegan_generator(sd, size, latent, noise, p=""):
    """Generator.forward(input_is_latent=True, return_features=True) with a (B, n_latent, 512) W+ code
    (reference e4e/models/stylegan2/model.py:475-552).  Returns (image, [feature after conv1 / each up-conv])."""
    B = latent.shape[0]
    log_size = int(math.log2(size))
    out = sd[p + "input.input"].repeat(B, 1, 1, 1)
    out = styled_conv(sd, p + "conv1.", out, latent[:, 0], noise[0])
    skip = to_rgb(sd, p + "to_rgb1.", out, latent[:, 1])
    feats = [out]
    i = 1
    for j in range(log_size - 2):
        out = styled_conv(sd, f"{p}convs.{2 * j}.", out, latent[:, i], noise[1 + 2 * j], mode="up")
        feats.append(out)
        out = styled_conv(sd, f"{p}convs.{2 * j + 1}.", out, latent[:, i + 1], noise[2 + 2 * j])
        skip = to_rgb(sd, f"{p}to_rgbs.{j}.", out, latent[:, i + 2], skip)
        i += 2
    return skip, feats


def stylegan_feats(sd, size, out_size, latent, noise, p=""):
    """E4e_embedding.get_stylegan_feats -> My_pSp.stylegan2_feat_forward (reference Loss/e4e_embedding.py:131-135,
    e4e/models/psp.py:235-248): features truncated to out_n_latent entries, image average-pooled to out_size."""
    img, feats = stylegan_generator(sd, size, latent, noise, p)
    out_n_latent = int(math.log2(out_size)) * 2 - 2
    return F.adaptive_avg_pool2d(img, (out_size, out_size)), feats[:out_n_latent]


# --------------------------------------------------------------------------------------------- e4e encoder (IR-SE50)
IR50_BLOCKS = ((64, 64, 3), (64, 128, 4), (128, 256, 14), (256, 512, 3))  # helpers.py:30-37


def ir_units():
    units = []
    for in_c, depth, n in IR50_BLOCKS:
        units.append((in_c, depth, 2))
        units += [(depth, depth, 1)] * (n - 1)
    return units


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"], False, 0.0, 1e-5)


def bottleneck_ir_se(sd, p, x, in_c, depth, stride):
    """reference e4e/models/encoders/helpers.py:89-113 (+ SEModule :57-73)."""
    if in_c == depth:
        shortcut = x[:, :, ::stride, ::stride]  # MaxPool2d(1, stride)
    else:
        shortcut = _bn(sd, p + "shortcut_layer.1.", F.conv2d(x, sd[p + "shortcut_layer.0.weight"], stride=stride))
    r = _bn(sd, p + "res_layer.0.", x)
    r = F.conv2d(r, sd[p + "res_layer.1.weight"], padding=1)
    r = F.prelu(r, sd[p + "res_layer.2.weight"])
    r = F.conv2d(r, sd[p + "res_layer.3.weight"], stride=stride, padding=1)
    r = _bn(sd, p + "res_layer.4.", r)
    g = r.mean(dim=(2, 3), keepdim=True)
    g = F.relu(F.conv2d(g, sd[p + "res_layer.5.fc1.weight"]))
    g = torch.sigmoid(F.conv2d(g, sd[p + "res_layer.5.fc2.weight"]))
    return r * g + shortcut


def gradual_style_block(sd, p, x, spatial):
    """reference e4e/models/encoders/psp_encoders.py:34-55: log2(spatial) x (3x3 stride-2 conv + LeakyReLU(0.01)),
    flatten, EqualLinear(lr_mul=1)."""
    for i in range(int(np.log2(spatial))):
        x = F.leaky_relu(F.conv2d(x, sd[f"{p}convs.{2 * i}.weight"], sd[f"{p}convs.{2 * i}.bias"], stride=2, padding=1), 0.01)
    x = x.reshape(-1, x.shape[1])
    return equal_linear(x, sd[p + "linear.weight"], sd[p + "linear.bias"])


def _upsample_add(x, y):
    return F.interpolate(x, size=y.shape[2:], mode="bilinear", align_corners=True) + y  # helpers.py:123-140


def encoder4editing(sd, x, p="", n_styles=18):
    """Encoder4Editing.forward at ProgressiveStage.Inference (reference psp_encoders.py:173-200)."""
    x = F.conv2d(x, sd[p + "input_layer.0.weight"], padding=1)
    x = F.prelu(_bn(sd, p + "input_layer.1.", x), sd[p + "input_layer.2.weight"])
    taps = {}
    for i, (in_c, depth, stride) in enumerate(ir_units()):
        x = bottleneck_ir_se(sd, f"{p}body.{i}.", x, in_c, depth, stride)
        if i in (6, 20, 23):
            taps[i] = x
    c1, c2, c3 = taps[6], taps[20], taps[23]
    w0 = gradual_style_block(sd, p + "styles.0.", c3, 16)
    w = w0.unsqueeze(1).repeat(1, n_styles, 1)
    feats, spatial = c3, 16
    for i in range(1, n_styles):
        if i == 3:
            feats = _upsample_add(c3, F.conv2d(c2, sd[p + "latlayer1.weight"], sd[p + "latlayer1.bias"]))
            p2, spatial = feats, 32
        elif i == 7:
            feats = _upsample_add(p2, F.conv2d(c1, sd[p + "latlayer2.weight"], sd[p + "latlayer2.bias"]))
            spatial = 64
        w[:, i] = w[:, i] + gradual_style_block(sd, f"{p}styles.{i}.", feats, spatial)
    return w


def get_w_plus(sd, img, latent_avg, p="encoder."):
    """E4e_embedding.get_w_plus -> My_pSp.forward (reference Loss/e4e_embedding.py:91-100, e4e/models/psp.py:145-165):
    bilinear resize to 256^2 (align_corners=False), encoder, + latent_avg, first 18 codes."""
    x = F.interpolate(img, (256, 256), mode="bilinear")
    codes = encoder4editing(sd, x, p)
    return (codes + latent_avg.unsqueeze(0))[:, :18]


# --------------------------------------------------------------------------------------------- Code_diffuser + DDPM
def _layer_norm(x, w=None, b=None):
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def _slrelu(x):
    return F.leaky_relu(x, 0.2) * math.sqrt(2)  # ScaledLeakyReLU, models/CodeDiffuser.py:50-59


def tacc_block(sd, p, x, embd, step):
    """TACC_block.forward + spatial_attention.forward (reference models/CodeDiffuser.py:86-116, :35-47)."""
    x = pixel_norm(x)  # over dim 1 = the 18 tokens
    K = F.linear(x, sd[p + "k_matrix.weight"])
    V = F.linear(x, sd[p + "v_matrix.weight"])
    c = torch.cat([embd, step], dim=-1)
    Q = F.linear(c, sd[p + "q_matrix.weight"]).permute(0, 2, 1)
    score = F.softmax(torch.matmul(K, Q) / math.sqrt(18), dim=-1)
    h = torch.matmul(score, V)
    q2 = F.linear(x, sd[p + "attention_layer.q_matrix.weight"])
    k2 = F.linear(c, sd[p + "attention_layer.k_matrix.weight"]).permute(0, 2, 1)
    v2 = F.linear(x, sd[p + "attention_layer.v_matrix.weight"])
    att = F.softmax(torch.matmul(k2, q2) / math.sqrt(x.shape[-1]), dim=1)
    t = _layer_norm(torch.matmul(v2, att))
    h = _layer_norm(h + t)

    def head(name, last):
        g = F.linear(c, sd[f"{p}{name}.0.weight"], sd[f"{p}{name}.0.bias"])
        g = _slrelu(_layer_norm(g, sd[f"{p}{name}.1.weight"], sd[f"{p}{name}.1.bias"]))
        return last(F.linear(g, sd[f"{p}{name}.3.weight"], sd[f"{p}{name}.3.bias"]))

    gamma = head("gamma_", torch.sigmoid)
    beta = head("beta_", _slrelu)
    return h * (1.0 + gamma) + beta


def code_diffuser(sd, x, embd, t, max_period, p="att_mapper."):
    """Code_diffuser.forward (reference models/CodeDiffuser.py:133-140): t/max_period appended as feature 513."""
    step = (t.float() / max_period).view(-1, 1, 1).repeat(1, embd.shape[1], 1)
    for i in range(4):
        x = tacc_block(sd, f"{p}{i}.", x, embd, step)
    return x


def ddpm_schedule(timesteps, linear_start=1e-4, linear_end=2e-2):
    """make_beta_schedule('linear') + register_schedule (reference ldm/util2.py:21-25, ldm/ddpm.py:288-328): float64
    numpy, stored as float32.  Returns (betas, alphas_cumprod, posterior_mean_coef1, posterior_mean_coef2)."""
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, timesteps, dtype=torch.float64).numpy() ** 2
    alphas = 1.0 - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1.0, ac[:-1])
    c1 = betas * np.sqrt(ac_prev) / (1.0 - ac)
    c2 = (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)
    f32 = lambda a: torch.tensor(a, dtype=torch.float32)  # noqa: E731
    return f32(betas), f32(ac), f32(c1), f32(c2)


def ddpm_sample(sd, cond, x_T, timesteps, linear_start=1e-4, linear_end=2e-2):
    """My_DDPM.forward(training=False) (reference ldm/ddpm.py:419-429 -> p_sample :370-376 -> p_mean_variance
    :357-368 -> q_posterior :348-352): x0-parameterised, the posterior MEAN is returned (no noise added, no clipping)."""
    _, _, c1, c2 = ddpm_schedule(timesteps, linear_start, linear_end)
    x = x_T
    B = cond.shape[0]
    for i in reversed(range(timesteps)):
        t = torch.full((B,), i, dtype=torch.long)
        x0 = code_diffuser(sd, x, cond, t, timesteps)
        x = c1[i] * x0 + c2[i] * x
    return x


# --------------------------------------------------------------------------------------------- whole path
def save_image_quantize(x):
    """torchvision.utils.save_image(normalize=True, range=(-1,1)) quantiser restated (reference restoration_test.py:
    138-157; torchvision 0.13 utils.py): clamp to [-1,1], map to [0,1], *255 + 0.5, clamp, uint8."""
    y = (x.clamp(-1, 1) + 1) / 2
    return (y * 255 + 0.5).clamp(0, 255).to(torch.uint8)


# --------------------------------------------------------------------------------------------- DDIM (BASELINE config 3)
def ddim_schedule(alphas_cumprod, S, eta=0.0):
    """make_ddim_timesteps('uniform') + make_ddim_sampling_parameters (reference ldm/util2.py:46-74): steps
    range(0, T, T // S) + 1; a_t = acp[steps]; a_prev = [acp[0]] + acp[steps[:-1]]; sigma = eta * sqrt(...)."""
    ac = alphas_cumprod.double().numpy()
    T = ac.shape[0]
    steps = np.asarray(list(range(0, T, T // S))) + 1
    assert steps.max() < T, "the reference indexes alphas_cumprod out of range when S == T (SURVEY section 0)"
    a = ac[steps]
    a_prev = np.asarray([ac[0]] + ac[steps[:-1]].tolist())
    sig = eta * np.sqrt((1 - a_prev) / (1 - a) * (1 - a / a_prev))
    return steps, a, a_prev, sig


def ddim_sample(sd, cond, x_T, timesteps, S, linear_start=1e-4, linear_end=2e-2):
    """DDIMSampler.sample(eta=0) (reference ldm/ddim.py:55-206) on (B, 18, 512) latents with the (x, t, c) ->
    Code_diffuser(x, c, t) adapter; like the reference's code it treats the network output as e_t."""
    _, ac, _, _ = ddpm_schedule(timesteps, linear_start, linear_end)
    steps, a, a_prev, sig = ddim_schedule(ac, S)
    x = x_T
    B = cond.shape[0]
    for index in reversed(range(len(steps))):
        t = torch.full((B,), int(steps[index]), dtype=torch.long)
        e_t = code_diffuser(sd, x, cond, t, timesteps)
        at = torch.tensor(a[index], dtype=torch.float32)
        ap = torch.tensor(a_prev[index], dtype=torch.float32)
        st = torch.tensor(sig[index], dtype=torch.float32)
        som = torch.tensor(np.sqrt(1.0 - a[index]), dtype=torch.float32)
        pred_x0 = (x - som * e_t) / at.sqrt()
        x = ap.sqrt() * pred_x0 + (1.0 - ap - st ** 2).sqrt() * e_t  # sigma = 0: the noise term vanishes
    return x
