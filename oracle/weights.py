"""Name-keyed synthetic checkpoints (TEST INFRASTRUCTURE).

`synth_state_dict(model, spec, seed)` fills every entry of a state-dict spec [(name, shape, dtype), ...] from
oracle.keyed_rng, choosing the distribution from the entry's role so that activations stay O(1) through all four
networks (SURVEY.md section 7 "fixture-init recipe"): the same call regenerates identical tensors in the build container
(fed to the real reference by tools/make_golden.py) and on the GPU box (fed to the oracle and to the HIP modules).
The specs themselves are data recorded from the reference's modules: tests/golden/state_specs.json.
"""
import json
import math
import os
import re

import torch

from . import keyed_rng as R
from .ops import make_kernel

SPEC_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "state_specs.json")


def load_specs():
    with open(SPEC_PATH) as f:
        return json.load(f)


def _fan_in(shape):
    n = 1
    for d in shape[1:]:
        n *= d
    return max(n, 1)


def synth_tensor(model, name, shape, dtype, seed):
    key = f"{model}/{name}"
    shape = tuple(shape)
    if dtype == "int64":
        return torch.zeros(shape, dtype=torch.int64)
    N = lambda s=1.0: R.normal(seed, key, shape) * s          # noqa: E731
    U = lambda lo, hi: R.uniform(seed, key, shape, lo, hi)    # noqa: E731
    leaf = name.rsplit(".", 1)[-1]

    # ---- buffers with fixed values
    if name.endswith("blur.kernel") or name.endswith("upsample.kernel"):
        k = make_kernel([1, 3, 3, 1])
        down = "encoder_convs" in name  # StyledConv_down blurs are not gain-compensated (models/RestoreNet.py:451-457)
        return k if down else k * 4
    if model == "discriminator" and leaf == "kernel":   # the Blur of every down-sampling ConvLayer (models/RestoreNet.py:1150-1156)
        return make_kernel([1, 3, 3, 1])
    if name.startswith("noises.") or ".noises." in name:
        return N()
    if name == "latent_avg":
        return N(0.1)

    # ---- e4e encoder (plain nn.Conv2d / BatchNorm2d / PReLU)
    if model == "e4e_encoder":
        if leaf == "running_mean":
            return N(0.1)
        if leaf == "running_var":
            return U(0.5, 1.5)
        is_bn = re.search(r"(input_layer\.1|res_layer\.0|res_layer\.4|shortcut_layer\.1)\.(weight|bias)$", name)
        if is_bn:
            return U(0.5, 1.5) if leaf == "weight" else N(0.1)
        if re.search(r"(input_layer\.2|res_layer\.2)\.weight$", name):  # PReLU slopes
            return U(0.15, 0.35)
        if name.endswith("linear.weight"):  # EqualLinear(lr_mul=1): N(0,1), scaled by 1/sqrt(in) in forward
            return N()
        if leaf == "bias":
            return N(0.1)
        return N(1.0 / math.sqrt(_fan_in(shape)))

    # ---- loss networks (torchvision architectures, oracle/tv_models.py): He-init convs keep the ReLU activations O(1)
    if model == "lpips_vgg":
        return N(0.1) if leaf == "bias" else N(math.sqrt(2.0 / _fan_in(shape)))
    if model == "arcface_resnet101":
        if leaf == "running_mean":
            return N(0.1)
        if leaf == "running_var":
            return U(0.5, 1.5)
        if re.search(r"(bn\d|downsample\.1)\.(weight|bias)$", name):
            if leaf == "bias":
                return N(0.1)
            # the last BatchNorm of a residual branch at 0.1-0.3: 33 stacked blocks keep the trunk O(1) instead of growing 6x
            return U(0.1, 0.3) if ".bn3." in name else U(0.5, 1.5)
        if name == "fc.bias":
            return N(0.1)
        if name == "fc.weight":
            return N(1.0 / math.sqrt(_fan_in(shape)))
        return N(math.sqrt(2.0 / _fan_in(shape)))

    # ---- Code_diffuser (nn.Linear / LayerNorm)
    if model == "diffuser":
        if re.search(r"(gamma_|beta_)\.1\.(weight|bias)$", name):  # LayerNorm affine
            return U(0.5, 1.5) if leaf == "weight" else N(0.1)
        if re.search(r"beta_\.3\.(weight|bias)$", name):
            # Conditioning of the sampler chain x <- c1*f(x, c) + c2*x (ldm/ddpm.py:400-429).  Every TACC block sees x
            # only through PixelNorm over the 18 tokens (models/CodeDiffuser.py:86), so its Jacobian w.r.t. x is
            # L_block / rms(x) with L_block ~ 2-4 for random projections, and rms(x) is set by the previous block's
            # output LN(.)*(1+gamma) + beta.  With the plain fan-in init the x-independent shift beta is O(1), the gain per
            # block is ~1.6 and T steps amplify a 1e-5 perturbation ~40x (T=4) to ~600x (T=50): the reference's own fp32
            # run is then 5e-4 ... 5e-3 away from its fp64 evaluation and no fp32 implementation can be pinned to 1e-3.
            # A trained denoiser contracts.  Scaling the output layer of the beta head by 4 (shapes and keys untouched) makes
            # the synthetic one contractive for BOTH kinds of condition the tests use: independent N(0,1) tokens (gain w.r.t.
            # x_T 4e-3 at T=4) and the e4e encoder's codes, whose 18 tokens share the w0 component (psp_encoders.py:181-198:
            # gain 0.11 at T=4, 0.01 at T=50; with x2 that case still expands 18x).  The reference's fp32 run is then
            # 3-5e-5 from its fp64 evaluation on latents of |max| ~25 -- which is what ONE TACC block evaluated in fp32 is
            # from fp64 on the same input, i.e. the chain no longer amplifies.  tools/condition_probe.py prints these
            # figures from the reference itself.
            return (N(0.1) if leaf == "bias" else N(1.0 / math.sqrt(_fan_in(shape)))) * 4.0
        if leaf == "bias":
            return N(0.1)
        w = N(1.0 / math.sqrt(_fan_in(shape)))
        if re.search(r"att_mapper\.\d+\.(q_matrix|k_matrix)\.weight$", name):
            # token-attention logits K.Q/sqrt(18) sum 512 products: with unit-variance projections they have std ~5 and
            # the 18-way softmax is one-hot, which makes the T-step chain chaotic in fp32 (the reference's own CPU fp32
            # run then differs from an fp64 run by O(1) at T=10).  Quarter-scale q/k keeps the logits O(0.3).
            w = w * 0.25
        return w

    # ---- StyleGAN2-style networks: Restoration_net, e4e decoder (Equal* layers: N(0,1)/lr_mul)
    if name.endswith("modulation.bias"):
        return torch.ones(shape)
    if name.endswith("noise.weight"):
        return U(0.05, 0.15)
    m = re.match(r"(decoder\.)?style\.(\d+)\.(weight|bias)$", name)
    if m:  # mapping network, lr_mul = 0.01
        return N(100.0) if leaf == "weight" else N(10.0)
    if name.endswith("input.input"):
        return N()
    if leaf == "bias":
        return N(0.1)
    w = N()
    if name.endswith("fusion.0.weight") and model == "restorenet":
        w = w * 0.5
    if re.search(r"to_rgbs?\d*\.(\d+\.)?conv\.weight$", name):
        w = w * 0.15  # ToRGB is not demodulated: its output scales with the latent's rms (~4.4 after the sampler chain)
    return w


def synth_state_dict(model, spec, seed=0, prefix=""):
    return {prefix + n: synth_tensor(model, n, s, d, seed) for n, s, d in spec}
