"""Deterministic parity cases shared by the golden generator (reference run), the oracle tests (CPU) and the HIP
parity tests (GPU).  TEST INFRASTRUCTURE.

A case is a name plus a dict of named fp32 input tensors drawn from oracle.keyed_rng -- nothing but the *outputs* of the
real reference has to be stored in tests/golden/.
"""
import torch

from . import keyed_rng as R
from .ops import make_kernel

SEED = 20251001

# ---------------------------------------------------------------------------------------------------- operator cases
# name -> (x shape, bias shape or None, slope, scale)
LRELU_CASES = {
    "lrelu_4d": ((2, 5, 7, 9), (5,)),
    "lrelu_2d": ((3, 8), (8,)),
    "lrelu_nobias": ((2, 3, 4, 4), None),
    "lrelu_vec4": ((2, 6, 8, 8), (6,)),
}

# name -> (x shape, kernel spec, up (x, y), down (x, y), pad (x0, x1, y0, y1))
FIR_CASES = {
    "fir_blur_after_up": ((2, 3, 9, 9), "blur4", (1, 1), (1, 1), (1, 1, 1, 1)),       # Blur pad (1,1): (2H+1) -> 2H
    "fir_blur_before_down": ((2, 3, 8, 8), "blur1", (1, 1), (1, 1), (2, 2, 2, 2)),    # Blur pad (2,2): H -> H+1
    "fir_upsample_rgb": ((1, 3, 8, 8), "blur4", (2, 2), (1, 1), (2, 1, 2, 1)),        # Upsample: up 2, pad (2,1)
    "fir_downsample": ((1, 2, 16, 16), "blur1", (1, 1), (2, 2), (1, 1, 1, 1)),
    "fir_generic_asym": ((1, 2, 7, 10), "rand3x5", (2, 3), (3, 2), (2, 1, 0, 3)),
    "fir_crop": ((1, 2, 12, 11), "rand4x4", (1, 1), (1, 1), (-1, 2, 1, -2)),
    "fir_tiles_129": ((1, 2, 129, 129), "rand4x4", (1, 1), (1, 1), (1, 1, 1, 1)),     # several 32x64 tiles + ragged edge
    "fir_tiles_k3": ((1, 2, 40, 70), "rand3x3", (1, 1), (1, 1), (1, 1, 1, 1)),
}


def fir_kernel(spec, name):
    if spec == "blur4":
        return make_kernel([1, 3, 3, 1]) * 4
    if spec == "blur1":
        return make_kernel([1, 3, 3, 1])
    kh, kw = (int(v) for v in spec[4:].split("x"))
    return R.normal(SEED, f"{name}/kernel", (kh, kw)) * 0.3


def lrelu_inputs(name):
    xs, bs = LRELU_CASES[name]
    x = R.normal(SEED, f"{name}/x", xs)
    b = R.normal(SEED, f"{name}/b", bs) if bs else None
    return x, b


def fir_inputs(name):
    xs, kspec, up, down, pad = FIR_CASES[name]
    return R.normal(SEED, f"{name}/x", xs), fir_kernel(kspec, name), up, down, pad


# ---------------------------------------------------------------------------------------------------- conv-layer cases
# modulated convs: name -> (kind, cin, cout, ksize, style_dim, x shape, extra)
MODCONV_CASES = {
    "modconv_same": ("same", 8, 12, 3, 32, (2, 8, 10, 10), {}),
    "modconv_up": ("up", 8, 8, 3, 32, (2, 8, 8, 8), {}),
    "modconv_down": ("down", 8, 16, 3, 32, (2, 8, 16, 16), {}),
    "modconv_rgb": ("same", 8, 3, 1, 32, (2, 8, 12, 12), {"demodulate": False}),
    "modconv_same_big": ("same", 32, 64, 3, 32, (1, 32, 40, 40), {}),
}
DILCONV_CASES = {f"dilconv_d{d}": (8, 4, (2, 8, 20, 20), d) for d in (1, 2, 4, 8)}
SMART_CASES = {"smart_16": (16, 16, 24, (2, 16, 12, 12))}
LARGECONV_CASES = {"largeconv_k1": (3, 16, 1, (2, 3, 8, 8)), "largeconv_k3": (16, 16, 3, (2, 16, 4, 4))}


def tensor(case, name, shape, scale=1.0):
    return R.normal(SEED, f"{case}/{name}", shape) * scale


def module_weights(case, named_shapes, kind="restorenet"):
    """Weights for a reference sub-module of a layer case: {param name: tensor} via the role-based synthesiser."""
    from .weights import synth_tensor
    return {n: synth_tensor(kind, n, tuple(s), "float32", _case_seed(case)) for n, s in named_shapes}


def _case_seed(case):
    return SEED + (sum(ord(c) for c in case) % 1000)


# ---------------------------------------------------------------------------------------------------- network cases
DIFFUSER_CASES = {
    # name -> (batch, timesteps, linear_start, linear_end)
    "ddpm_T4": (2, 4, 0.1, 0.99),          # what restoration_test.py:35-38 runs
    "ddpm_T10": (2, 10, 1e-4, 2e-2),
    "ddpm_T50": (2, 50, 1e-4, 2e-2),       # BASELINE configs[1]: the full-length chain, default betas (ldm/ddpm.py:256-257)
}
DDIM_CASE = ("ddim_T50_S25", 2, 50, 25)  # name, batch, DDPM steps, DDIM steps (BASELINE config 3; default betas)


def diffuser_inputs(name):
    B = DIFFUSER_CASES[name][0] if name in DIFFUSER_CASES else DDIM_CASE[1]
    cond = tensor(name, "cond", (B, 18, 512))
    x_T = tensor(name, "x_T", (B, 18, 512))
    return cond, x_T


def image_batch(case, B, size):
    return R.uniform(SEED, f"{case}/lq", (B, 3, size, size), -1.0, 1.0)


def noise_list(case, tag, shapes):
    return [R.normal(SEED, f"{case}/{tag}{i}", s) for i, s in enumerate(shapes)]


def feat_sample(f):
    """Large feature maps are stored in the fixtures as a channel-strided sample."""
    return f[:, ::16] if f[0].numel() > 65536 else f
