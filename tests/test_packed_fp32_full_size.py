"""Full-size bit-stability fence for the kernels that mix packed-fp32 arithmetic, MFMAs, LDS-DMA with hand-counted waits and persistent
workgroups (review r5, weak 1.i): the round-4 miscompare of conv_bf16_rv.hip hit 0.01 - 0.05 % of the outputs, "never the same
ones", and only when every CU walked several items -- the small-map repeat tests of tests/test_wino4f.py / test_wino_rs.py do not fence that.
Here every launch runs at the size the headline step runs it (batch 8, every CU walks many items), 50 launches each, ALL outputs bit-identical
across launches, images 0 and 7 against float64 F.conv2d.  `pytest -m gpu`."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = "cuda"
LAUNCHES = 50


def _fp64(x, ws, dils, s_in, demod, bias, nz, nw, res):
    xd = x.double() * s_in.double().view(x.shape[0], -1, 1, 1)
    y = torch.cat([F.conv2d(xd, w.double(), padding=d, dilation=d) for w, d in zip(ws, dils)], dim=1)
    y = y * demod.double().view(x.shape[0], -1, 1, 1) + nz.double() * nw
    return F.leaky_relu(y + bias.double().view(1, -1, 1, 1), 0.2) * math.sqrt(2.0) + res.double()


@pytest.mark.parametrize("name,B,Cin,Cg,S,dils,how,tol", [
    ("conv_wino4f_kernel", 8, 64, 64, 512, (1,), dict(winograd=5), 6e-5),                                  # 64 -> 64 at 512^2
    ("conv_wino4f_groups_kernel", 8, 128, 32, 256, (1, 2, 4, 8), dict(winograd=5), 6e-5),                  # 128 -> 4 x 32 at 256^2
    ("conv_wino4f_groups_kernel", 8, 512, 128, 64, (1, 2, 4, 8), dict(winograd=5), 1.2e-4),                # 512 -> 4 x 128 at 64^2
    ("conv_wino_rs_kernel", 8, 64, 16, 512, (1, 2, 4, 8), dict(winograd=True, wino_form=3), 2e-5),         # 64 -> 4 x 16 at 512^2
])
def test_packed_fp32_kernels_bit_stable_full_size(name, B, Cin, Cg, S, dils, how, tol):
    from vspbfr_amd import hip_ops as H
    G = len(dils)
    g_ = torch.Generator(device=DEV).manual_seed(Cin * 7 + S)
    rnd = lambda *s: torch.randn(*s, generator=g_, device=DEV)                                    # noqa: E731
    x = rnd(B, Cin, S, S)
    ws = [rnd(Cg, Cin, 3, 3) / math.sqrt(Cin * 9) for _ in dils]
    s_in = torch.rand(B, Cin, generator=g_, device=DEV) + 0.5
    demod = torch.rand(B, G * Cg, generator=g_, device=DEV) + 0.5
    bias, nz, res = rnd(G * Cg), rnd(B, 1, S, S), rnd(B, G * Cg, S, S)
    nw = torch.tensor([0.3], device=DEV)
    wp = torch.stack([H.pack_weight(w_)[0] for w_ in ws]).contiguous() if G > 1 else H.pack_weight(ws[0])
    pc = H.PackedConv(wp, G, Cg, Cin, 3, 3, 1, dils, dils)
    kw = dict(in_scale=s_in, out_scale=demod, noise=nz, noise_w=nw, act2=1, bias2=bias, res1=res, **how)
    prof = H.ConvProfiler()
    H.PROFILER = prof
    try:
        y0 = H.conv2d_packed(x, pc, **kw)
    finally:
        H.PROFILER = None
    want = {"conv_wino4f_kernel": "wino4f", "conv_wino4f_groups_kernel": "wino4f", "conv_wino_rs_kernel": "wino"}[name]
    kinds = sorted(r[3][7] for r in prof.records)
    assert kinds == [want], (name, kinds)
    out = torch.empty_like(y0)
    bad = 0
    for i in range(LAUNCHES - 1):
        H.conv2d_packed(x, pc, out=out, **kw)
        if not torch.equal(out, y0):
            bad += 1
            n = int((out != y0).sum())
            print(f"{name}: launch {i + 1} differs on {n} of {y0.numel()} outputs ({100.0 * n / y0.numel():.4f} %)")
        out.fill_(float("nan"))
    assert bad == 0, f"{name}: {bad} of {LAUNCHES - 1} repeat launches differ from the first"
    for b in (0, B - 1):
        ref = _fp64(x[b:b + 1].cpu(), [w_.cpu() for w_ in ws], dils, s_in[b:b + 1].cpu(), demod[b:b + 1].cpu(), bias.cpu(), nz[b:b + 1].cpu(),
                    0.3, res[b:b + 1].cpu()).numpy()
        got = y0[b:b + 1].cpu().double().numpy()
        err, lim = np.abs(got - ref).max(), tol * (1.0 + np.abs(ref).max())
        assert np.isfinite(got).all() and err <= lim, f"{name} image {b}: max|d| = {err:.3e} > {lim:.3e}"
