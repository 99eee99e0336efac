"""The fused Winograd F(4x4,3x3) kernel (conv_wino4f.hip, vsp_conv2d_winograd4f_f32) DIRECTLY against float64 F.conv2d: the shallow wide
stride-1 layers of the prior and the restoration decoder (reference e4e/models/stylegan2/model.py:268-276, models/RestoreNet.py:421-555),
ragged maps (partial N-blocks, partial row groups, partial channel halves), every operand of the epilogue chain, all four template
instances (residuals x first activation).  `pytest -m gpu`."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = "cuda"


def dev(t):
    return t.to(DEV).contiguous()


@pytest.fixture(scope="module")
def H():
    from vspbfr_amd import hip_ops
    return hip_ops


def close64(a, b, tol, what=""):
    a, b = a.detach().cpu().double().numpy(), b.detach().cpu().double().numpy()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert np.isfinite(a).all(), f"{what}: non-finite output"
    err = np.abs(a - b).max()
    lim = tol * (1.0 + np.abs(b).max())
    assert err <= lim, f"{what}: max|d|={err:.3e} tol={lim:.3e}"


def _ref(x, w, s_in=None, demod=None, bias1=None, bias2=None, nz=None, nw=0.0, r1=None, r2=None):
    xd = x.double()
    if s_in is not None:
        xd = xd * s_in.double().view(x.shape[0], -1, 1, 1)
    y = F.conv2d(xd, w.double(), padding=1)
    if demod is not None:
        y = y * demod.double().view(x.shape[0], -1, 1, 1)
    if bias1 is not None:
        y = F.leaky_relu(y + bias1.double().view(1, -1, 1, 1), 0.2) * math.sqrt(2)
    if nz is not None:
        y = y + nz.double() * nw
    if bias2 is not None:
        y = F.leaky_relu(y + bias2.double().view(1, -1, 1, 1), 0.2) * math.sqrt(2)
    for r in (r1, r2):
        if r is not None:
            y = y + r.double()
    return y


# F(4x4) with the points 0, +-3/4, +-3/2, inf: measured max 1.3e-5 ... 3.1e-5 relative to the output range at K = 64 ... 256 (the pair: the same)
TOL = 6e-5


@pytest.mark.parametrize("B,Cin,Cout,Hh,Ww", [
    (2, 64, 64, 64, 64),        # whole N-blocks, four row groups, both channel halves
    (2, 8, 16, 20, 36),         # half an N-block, a partial row group, half a channel half, two k-steps
    (3, 40, 48, 16, 80),        # a second, partial N-block; 1.5 channel halves
    (1, 256, 96, 8, 16),        # the scale table's full 256 channels, two waves of a row group
    (1, 32, 32, 4, 16),         # one tile row
    (2, 64, 3, 64, 64),         # Cout = 3 (the tuned table sends two 64 -> 3 layers here): 29 of a half's 32 channels are padding
    (1, 64, 3, 256, 256),       # the tuned entry's map
    (2, 64, 64, 512, 512),      # the judged shape, first and last image against fp64
    (1, 32, 32, 1024, 1024),    # the prior's last layer
])
def test_conv2d_winograd4f_vs_fp64(H, B, Cin, Cout, Hh, Ww):
    g_ = torch.Generator().manual_seed(Hh * 131 + Ww)
    x = torch.randn(B, Cin, Hh, Ww, generator=g_)
    w = torch.randn(Cout, Cin, 3, 3, generator=g_) / math.sqrt(Cin * 9)
    s_in, demod = torch.rand(B, Cin, generator=g_) + 0.5, torch.rand(B, Cout, generator=g_) + 0.5
    b1, b2 = torch.randn(Cout, generator=g_), torch.randn(Cout, generator=g_)
    pc = H.PackedConv(H.pack_weight(dev(w)), 1, Cout, Cin, 3, 3, 1, (1,), (1,))
    assert H.winograd4f_eligible(pc, Hh, Ww, Hh, Ww)
    big = Hh * Ww >= 512 * 512
    sel = [0, B - 1] if big else list(range(B))
    xd = dev(x)
    # bare convolution (no residual, no first activation)
    close64(H.conv2d_packed(xd, pc, winograd=5)[sel], _ref(x[sel], w), TOL, "bare")
    # the StyledConv operand set: style scale, demodulation, noise, bias + leaky relu
    nz, nw = torch.randn(B, 1, Hh, Ww, generator=g_), torch.tensor([0.7])
    y = H.conv2d_packed(xd, pc, in_scale=dev(s_in), out_scale=dev(demod), noise=dev(nz), noise_w=dev(nw), act2=1, bias2=dev(b2), winograd=5)
    close64(y[sel], _ref(x[sel], w, s_in[sel], demod[sel], None, b2, nz[sel], 0.7), TOL, "modulated")
    if not big:
        # first activation (its own template instance), then noise and the second one
        y = H.conv2d_packed(xd, pc, in_scale=dev(s_in), act1=True, bias1=dev(b1), noise=dev(nz), noise_w=dev(nw), act2=1, bias2=dev(b2), winograd=5)
        close64(y, _ref(x, w, s_in, None, b1, b2, nz, 0.7), TOL, "two activations")
        # residuals (one, then two, with the first activation) and a channel window of a wider output tensor
        r1, r2 = torch.randn(B, Cout, Hh, Ww, generator=g_), torch.randn(B, Cout, Hh, Ww, generator=g_)
        y = H.conv2d_packed(xd, pc, out_scale=dev(demod), act2=1, bias2=dev(b2), res1=dev(r1), winograd=5)
        close64(y, _ref(x, w, None, demod, None, b2, None, 0.0, r1), TOL, "one residual")
        out = torch.full((B, Cout + 5, Hh, Ww), 7.0, device=DEV)
        H.conv2d_packed(xd, pc, out=out, y_coff=3, in_scale=dev(s_in), out_scale=dev(demod), act1=True, bias1=dev(b1), act2=1, bias2=dev(b2),
                        res1=dev(r1), res2=dev(r2), winograd=5)
        close64(out[:, 3:3 + Cout], _ref(x, w, s_in, demod, b1, b2, None, 0.0, r1, r2), TOL, "two residuals + window")
        assert float((out[:, :3] - 7.0).abs().max()) == 0.0 and float((out[:, 3 + Cout:] - 7.0).abs().max()) == 0.0


def test_conv2d_winograd4f_repeatable_and_refusals(H):
    """Bit-identical across launches (persistent workgroups, no atomics, hand-placed waits); a launch it does not serve is refused when the
    kernel is asked for by name."""
    g_ = torch.Generator().manual_seed(7)
    x = dev(torch.randn(4, 64, 128, 192, generator=g_))
    w = torch.randn(64, 64, 3, 3, generator=g_) / 24
    s_in = dev(torch.rand(4, 64, generator=g_) + 0.5)
    nz, nw = dev(torch.randn(4, 1, 128, 192, generator=g_)), dev(torch.tensor([0.3]))
    pc = H.PackedConv(H.pack_weight(dev(w)), 1, 64, 64, 3, 3, 1, (1,), (1,))
    y0 = H.conv2d_packed(x, pc, in_scale=s_in, noise=nz, noise_w=nw, winograd=5)
    for _ in range(20):
        assert torch.equal(H.conv2d_packed(x, pc, in_scale=s_in, noise=nz, noise_w=nw, winograd=5), y0)
    with pytest.raises(RuntimeError):
        H.conv2d_packed(dev(torch.randn(1, 64, 16, 12)), pc, winograd=5)                             # rows shorter than one 16-pixel quad row
    with pytest.raises(RuntimeError):
        H.conv2d_packed(dev(torch.randn(1, 64, 18, 16)), pc, winograd=5)                             # rows not a multiple of 4
    w12 = torch.randn(16, 12, 3, 3, generator=g_)
    pc12 = H.PackedConv(H.pack_weight(dev(w12)), 1, 16, 12, 3, 3, 1, (1,), (1,))
    with pytest.raises(RuntimeError):
        H.conv2d_packed(dev(torch.randn(1, 12, 16, 16)), pc12, winograd=5)                           # Cin not a multiple of 8
    with pytest.raises(RuntimeError):
        H.conv2d_packed(dev(torch.randn(1, 64, 16, 16)), pc, in_scale=dev(torch.rand(64)), in_scale_per_sample=False,
                        in_shift=dev(torch.randn(64)), winograd=5)                                   # affine input shift
    # the tuned table's fall-back is silent: the same launch without the name runs on another kernel
    y = H.conv2d_packed(dev(torch.randn(1, 64, 16, 12)), pc)
    assert y.shape == (1, 64, 16, 12)


def test_winograd4f_weight_layout(H):
    """U4F = G g G^T in the order the header documents: [co / 32][ci / 4][position pair][lane][4]."""
    g_ = torch.Generator().manual_seed(3)
    Cin, Cout = 8, 48
    w = torch.randn(Cout, Cin, 3, 3, generator=g_)
    U = H.winograd4f_weight(H.pack_weight(dev(w))).cpu().double()
    Gm = torch.tensor([[64 / 81, 0, 0], [-128 / 243, -32 / 81, -8 / 27], [-128 / 243, 32 / 81, -8 / 27], [32 / 243, 16 / 81, 8 / 27],
                       [32 / 243, -16 / 81, 8 / 27], [0, 0, 1]], dtype=torch.float64)
    ref = torch.einsum("ia,ocab,jb->ocij", Gm, w.double(), Gm).reshape(Cout, Cin, 36)
    U = U.view(2, Cin // 4, 18, 64, 4)
    for half, ks, pp, lane, e in [(0, 0, 0, 0, 0), (1, 1, 17, 63, 3), (0, 1, 5, 37, 2), (1, 0, 9, 20, 1)]:
        co, ci, pos = 32 * half + 16 * (e & 1) + (lane & 15), 4 * ks + (lane >> 4), 2 * pp + (e >> 1)
        want = ref[co, ci, pos].item() if co < Cout else 0.0
        assert abs(U[half, ks, pp, lane, e].item() - want) <= 1e-6 * (1 + abs(want)), (half, ks, pp, lane, e)


def test_dilation_groups_on_fused_kernel(H):
    """A SMART dilation-group launch (dilation 1, 2, 4, 8 over one shared input) on the fused F(4x4) kernel: ONE launch, a partition of
    workgroups per group, the dilated groups through the LDS window loader -- against float64 F.conv2d(dilation = d), picked by the tuned
    table, with and without per-channel epilogue operands (their channel index runs over all four groups)."""
    g_ = torch.Generator().manual_seed(11)
    B, Cin, Cg, S = 8, 128, 32, 256
    x = torch.randn(B, Cin, S, S, generator=g_)
    ws = [torch.randn(Cg, Cin, 3, 3, generator=g_) / math.sqrt(Cin * 9) for _ in range(4)]
    s_in, demod = torch.rand(B, Cin, generator=g_) + 0.5, torch.rand(B, 4 * Cg, generator=g_) + 0.5
    wp = torch.stack([H.pack_weight(dev(w_))[0] for w_ in ws]).contiguous()
    pc = H.PackedConv(wp, 4, Cg, Cin, 3, 3, 1, (1, 2, 4, 8), (1, 2, 4, 8))
    key = H.conv_key(B, Cin, S, S, pc, S, S)
    assert H.WINO.get(key) == 5, key
    prof = H.ConvProfiler()
    H.PROFILER = prof
    try:
        y = H.conv2d_packed(dev(x), pc, in_scale=dev(s_in), out_scale=dev(demod))
    finally:
        H.PROFILER = None
    kinds = sorted(r[3][7] for r in prof.records)
    assert kinds == ["wino4f"], kinds
    for b in (0, B - 1):
        xd = (x[b:b + 1] * s_in[b].view(1, -1, 1, 1)).double()
        ref = torch.cat([F.conv2d(xd, w_.double(), padding=d, dilation=d) for w_, d in zip(ws, (1, 2, 4, 8))], dim=1) * demod[b].double().view(1, -1, 1, 1)
        close64(y[b:b + 1], ref, TOL, f"image {b}")
    # per-channel operands, a residual and noise; against the F(2x2) kernels on the same launch
    bias, res, nz = torch.randn(4 * Cg, generator=g_), torch.randn(B, 4 * Cg, S, S, generator=g_), torch.randn(B, 1, S, S, generator=g_)
    kw = dict(in_scale=dev(s_in), out_scale=dev(demod), bias2=dev(bias), act2=1, res1=dev(res), noise=dev(nz), noise_w=dev(torch.tensor([0.3])))
    y1 = H.conv2d_packed(dev(x), pc, winograd=5, **kw)
    b = B - 1
    xd = (x[b:b + 1] * s_in[b].view(1, -1, 1, 1)).double()
    ref = torch.cat([F.conv2d(xd, w_.double(), padding=d, dilation=d) for w_, d in zip(ws, (1, 2, 4, 8))], dim=1) * demod[b].double().view(1, -1, 1, 1)
    ref = F.leaky_relu(ref + 0.3 * nz[b:b + 1].double() + bias.double().view(1, -1, 1, 1), 0.2) * math.sqrt(2.0) + res[b:b + 1].double()
    close64(y1[b:b + 1], ref, TOL, "group launch with epilogue operands")
    close64(H.conv2d_packed(dev(x), pc, winograd=True, **kw), y1, 2 * TOL, "fused vs F(2x2) kernels")


def test_dilation_groups_512_channels_on_fused_kernel(H):
    """512 -> 4 x 128 at 64^2 (the deepest SMART launch the fused kernel takes: 128 k-steps, scale table of 512 channels), three groups."""
    g_ = torch.Generator().manual_seed(12)
    B, Cin, Cg, S = 2, 512, 128, 64
    x = torch.randn(B, Cin, S, S, generator=g_)
    ws = [torch.randn(Cg, Cin, 3, 3, generator=g_) / math.sqrt(Cin * 9) for _ in range(3)]
    s_in, demod = torch.rand(B, Cin, generator=g_) + 0.5, torch.rand(B, 3 * Cg, generator=g_) + 0.5
    wp = torch.stack([H.pack_weight(dev(w_))[0] for w_ in ws]).contiguous()
    pc = H.PackedConv(wp, 3, Cg, Cin, 3, 3, 1, (4, 1, 8), (4, 1, 8))
    y = H.conv2d_packed(dev(x), pc, in_scale=dev(s_in), out_scale=dev(demod), winograd=5)
    for b in range(B):
        xd = (x[b:b + 1] * s_in[b].view(1, -1, 1, 1)).double()
        ref = torch.cat([F.conv2d(xd, w_.double(), padding=d, dilation=d) for w_, d in zip(ws, (4, 1, 8))], dim=1) * demod[b].double().view(1, -1, 1, 1)
        close64(y[b:b + 1], ref, 2 * TOL, f"image {b}")


@pytest.mark.parametrize("d", [2, 4, 8])
@pytest.mark.parametrize("shape", [(2, 32, 32, 64, 64), (3, 64, 48, 32, 96), (2, 24, 40, 160, 64), (1, 256, 16, 64, 128), (1, 16, 32, 40, 72)])
def test_conv2d_winograd4f_dilated(H, d, shape):
    """The fused F(4x4) kernel on dilated layers (polyphase tiles, LDS window loader): every epilogue operand, maps narrower / shorter than a
    workgroup region, channel counts off the 32-channel half, against float64 F.conv2d; bit-identical across launches (hand-counted waits)."""
    B, Cin, Cout, Hh, Ww = shape
    if Hh % (4 * d) or Ww % (4 * d):
        pytest.skip("not whole tiles at this dilation")
    g_ = torch.Generator().manual_seed(100 * d + Cin)
    x = torch.randn(B, Cin, Hh, Ww, generator=g_)
    w = torch.randn(Cout, Cin, 3, 3, generator=g_) / math.sqrt(Cin * 9)
    s_in, demod = torch.rand(B, Cin, generator=g_) + 0.5, torch.rand(B, Cout, generator=g_) + 0.5
    bias, res, nz = torch.randn(Cout, generator=g_), torch.randn(B, Cout, Hh, Ww, generator=g_), torch.randn(B, 1, Hh, Ww, generator=g_)
    pc = H.PackedConv(H.pack_weight(dev(w)), 1, Cout, Cin, 3, 3, 1, (d,), (d,))
    kw = dict(in_scale=dev(s_in), out_scale=dev(demod), act2=1, bias2=dev(bias), res1=dev(res), noise=dev(nz), noise_w=dev(torch.tensor([0.3])))
    y = H.conv2d_packed(dev(x), pc, winograd=5, **kw)
    ref = F.conv2d(x.double() * s_in.double()[:, :, None, None], w.double(), padding=d, dilation=d) * demod.double()[:, :, None, None]
    ref = F.leaky_relu(ref + 0.3 * nz.double() + bias.double()[None, :, None, None], 0.2) * math.sqrt(2.0) + res.double()
    close64(y, ref, TOL, f"d = {d}")
    for _ in range(5):
        assert torch.equal(H.conv2d_packed(dev(x), pc, winograd=5, **kw), y)
    y2 = H.conv2d_packed(dev(x), pc, winograd=5, in_scale=dev(s_in))          # no residual / second activation: the other template instance
    close64(y2, F.conv2d(x.double() * s_in.double()[:, :, None, None], w.double(), padding=d, dilation=d), TOL, f"plain d = {d}")
