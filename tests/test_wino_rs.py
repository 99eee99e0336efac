"""The register-resident-U Winograd kernel (conv_wino_rs.hip, vsp_conv2d_winograd_f32 with tile_hint = 3) DIRECTLY against float64
F.conv2d(dilation = d) -- not against another HIP kernel: the dilation groups of the SMART layers (reference models/RestoreNet.py:179-244,
270-418) and the plain low-channel layers, ragged maps, partial channel blocks, the whole epilogue chain.  `pytest -m gpu`."""
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


def _ref(x, ws, dils, s_in=None, demod=None, bias=None, nz=None, nw=0.0, r1=None, r2=None):
    xd = x.double()
    if s_in is not None:
        xd = xd * s_in.double().view(x.shape[0], -1, 1, 1)
    y = torch.cat([F.conv2d(xd, w_.double(), padding=d, dilation=d) for w_, d in zip(ws, dils)], dim=1)
    if demod is not None:
        y = y * demod.double().view(x.shape[0], -1, 1, 1)
    if nz is not None:
        y = y + nz.double() * nw
    if bias is not None:
        y = F.leaky_relu(y + bias.double().view(1, -1, 1, 1), 0.2) * math.sqrt(2)
    for r in (r1, r2):
        if r is not None:
            y = y + r.double()
    return y


@pytest.mark.parametrize("B,Cin,Cg,Hh,Ww,dils", [
    (2, 64, 16, 64, 64, (1, 2, 4, 8)),      # the SMART branch launch (64 -> 4 x 16), whole items
    (1, 64, 16, 128, 96, (1, 2, 4, 8)),     # several column blocks / row blocks per residue class
    (3, 40, 16, 37, 20, (1, 2, 4, 8)),      # ragged: rows not a multiple of 8 d, one partial column block, Cin not a multiple of 8
    (1, 24, 24, 19, 36, (2, 8)),            # partial second channel block, two groups
    (2, 32, 32, 48, 64, (1,)),              # plain layer: two blocks of 16 channels, four stages
    (1, 64, 64, 40, 32, (1,)),              # plain 64 -> 64
    (1, 16, 8, 9, 8, (4,)),                 # map smaller than one item
    (2, 64, 16, 512, 512, (1, 2, 4, 8)),    # the judged shape (64 -> 4 x 16 at 512^2), first and last image against fp64
])
def test_conv2d_winograd_rs_vs_fp64(H, B, Cin, Cg, Hh, Ww, dils):
    g_ = torch.Generator().manual_seed(Hh * 131 + Ww)
    G = len(dils)
    x = torch.randn(B, Cin, Hh, Ww, generator=g_)
    ws = [torch.randn(Cg, Cin, 3, 3, generator=g_) / math.sqrt(Cin * 9) for _ in dils]
    s_in, demod, bias = torch.rand(B, Cin, generator=g_) + 0.5, torch.rand(B, G * Cg, generator=g_) + 0.5, torch.randn(G * Cg, generator=g_)
    wp = torch.stack([H.pack_weight(dev(w_))[0] for w_ in ws]).contiguous()
    pc = H.PackedConv(wp, G, Cg, Cin, 3, 3, 1, dils, dils)
    big = Hh * Ww >= 512 * 512
    sel = [0, B - 1] if big else list(range(B))
    # bare convolution
    y = H.conv2d_packed(dev(x), pc, winograd=True, wino_form=3)
    close64(y[sel], _ref(x[sel], ws, dils), 2e-5, "bare")
    # modulated layer: style scale, demodulation, noise, bias + leaky relu
    nz, nw = torch.randn(B, 1, Hh, Ww, generator=g_), torch.tensor([0.7])
    y = H.conv2d_packed(dev(x), pc, in_scale=dev(s_in), out_scale=dev(demod), noise=dev(nz), noise_w=dev(nw), act2=1, bias2=dev(bias),
                        winograd=True, wino_form=3)
    close64(y[sel], _ref(x[sel], ws, dils, s_in[sel], demod[sel], bias, nz[sel], 0.7), 2e-5, "modulated")
    if not big:
        # residuals and a channel window of a wider output tensor
        r1, r2 = torch.randn(B, G * Cg, Hh, Ww, generator=g_), torch.randn(B, G * Cg, Hh, Ww, generator=g_)
        out = torch.full((B, G * Cg + 5, Hh, Ww), 7.0, device=DEV)
        H.conv2d_packed(dev(x), pc, out=out, y_coff=3, in_scale=dev(s_in), out_scale=dev(demod), act2=1, bias2=dev(bias), res1=dev(r1), res2=dev(r2),
                        winograd=True, wino_form=3)
        close64(out[:, 3:3 + G * Cg], _ref(x, ws, dils, s_in, demod, bias, None, 0.0, r1, r2), 2e-5, "residuals + window")
        assert float((out[:, :3] - 7.0).abs().max()) == 0.0 and float((out[:, 3 + G * Cg:] - 7.0).abs().max()) == 0.0


def test_conv2d_winograd_rs_repeatable_and_refusals(H):
    """Bit-identical across launches (persistent workgroups, no atomics); the named form refuses what it does not serve."""
    g_ = torch.Generator().manual_seed(5)
    x = dev(torch.randn(2, 64, 96, 96, generator=g_))
    ws = [torch.randn(16, 64, 3, 3, generator=g_) / 24 for _ in range(4)]
    wp = torch.stack([H.pack_weight(dev(w_))[0] for w_ in ws]).contiguous()
    pc = H.PackedConv(wp, 4, 16, 64, 3, 3, 1, (1, 2, 4, 8), (1, 2, 4, 8))
    y0 = H.conv2d_packed(x, pc, winograd=True, wino_form=3)
    for _ in range(10):
        assert torch.equal(H.conv2d_packed(x, pc, winograd=True, wino_form=3), y0)
    w128 = torch.randn(32, 128, 3, 3, generator=g_)
    pc128 = H.PackedConv(H.pack_weight(dev(w128)), 1, 32, 128, 3, 3, 1, (1,), (1,))
    with pytest.raises(RuntimeError):
        H.conv2d_packed(dev(torch.randn(1, 128, 16, 16)), pc128, winograd=True, wino_form=3)      # more than 64 input channels
    pc64 = H.PackedConv(H.pack_weight(dev(ws[0])), 1, 16, 64, 3, 3, 1, (1,), (1,))
    with pytest.raises(RuntimeError):
        H.conv2d_packed(dev(torch.randn(1, 64, 16, 18)), pc64, winograd=True, wino_form=3)        # rows are not whole 16-byte segments
    with pytest.raises(RuntimeError):
        H.conv2d_packed(dev(torch.randn(1, 64, 16, 16)), pc64, in_scale=dev(torch.rand(64)), in_scale_per_sample=False,
                        in_shift=dev(torch.randn(64)), winograd=True, wino_form=3)                 # affine input shift
