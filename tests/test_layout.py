"""Checkpoint-layout contract (CPU): the vspbfr_amd modules expose exactly the state-dict keys/shapes/dtypes recorded from
the reference's modules (tests/golden/state_specs.json, written by tools/make_golden.py), so the published checkpoints load
with strict=True."""
import json
import os
from argparse import Namespace

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SPECS = json.load(open(os.path.join(ROOT, "tests", "golden", "state_specs.json")))


def build(name):
    from vspbfr_amd.diffusion import Code_diffuser
    from vspbfr_amd.discriminator import Discriminator
    from vspbfr_amd.e4e import Encoder4Editing, Generator
    from vspbfr_amd.id_loss import ResNet101
    from vspbfr_amd.lpips import PNetLin
    from vspbfr_amd.restorenet import Restoration_net
    return {
        "lpips_vgg": lambda: PNetLin(),                      # my_lpips PNetLin(vgg) over torchvision vgg16().features
        "arcface_resnet101": lambda: ResNet101(256),         # torchvision resnet101(num_classes=256) of Loss/id_loss.py:13
        "restorenet512": lambda: Restoration_net(512, 512, 8, channel_multiplier=2),
        "restorenet64": lambda: Restoration_net(64, 512, 8),
        "diffuser": lambda: Code_diffuser(timesteps=4),
        "e4e_encoder": lambda: Encoder4Editing(50, "ir_se", Namespace(input_channel=3, stylegan_size=1024)),
        "e4e_decoder1024": lambda: Generator(1024, 512, 8, channel_multiplier=2),
        "e4e_decoder64": lambda: Generator(64, 512, 8, channel_multiplier=2),
        "discriminator512": lambda: Discriminator(512),
        "discriminator64": lambda: Discriminator(64),
    }[name]()


@pytest.mark.parametrize("name", sorted(SPECS))
def test_state_dict_layout(name):
    got = {k: (list(v.shape), str(v.dtype).replace("torch.", "")) for k, v in build(name).state_dict().items()}
    ref = {k: (s, d) for k, s, d in SPECS[name]}
    assert set(got) == set(ref), (sorted(set(ref) - set(got))[:5], sorted(set(got) - set(ref))[:5])
    assert all(got[k] == ref[k] for k in ref)


def test_ddpm_buffers_and_schedule():
    import numpy as np

    from vspbfr_amd.diffusion import Code_diffuser, My_DDPM
    d = My_DDPM(Code_diffuser(4), timesteps=4, linear_start=0.1, linear_end=0.99)
    bufs = [k for k in d.state_dict() if not k.startswith("model.")]
    assert bufs == ["betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
                    "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                    "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"]
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "diffuser.npz")))
    np.testing.assert_array_equal(d.posterior_mean_coef1.numpy(), g["ddpm_T4/coef1"])  # bit-exact vs the reference's buffers
    np.testing.assert_array_equal(d.posterior_mean_coef2.numpy(), g["ddpm_T4/coef2"])


def test_shard_range_covers_batch():
    from vspbfr_amd.pipeline import shard_range
    for n in (0, 1, 7, 8, 128, 131):
        for w in (1, 2, 4, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.gpu
def test_winograd_weight_fragment_layout():
    """The Winograd weight transform + fragment ordering (hip_ops.winograd_weight -> vsp_winograd_weight_f32) against a numpy
    F(2x2,3x3) evaluation that reads U back through the documented index formula of vsp_conv2d_winograd_f32."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    from vspbfr_amd import hip_ops as H
    from vspbfr_amd._lib import lib
    rng = np.random.default_rng(5)
    for G, cin, cout_g, dils in ((1, 6, 20, (1,)), (4, 5, 16, (1, 2, 4, 8)), (1, 9, 70, (1,))):
        w = rng.standard_normal((G, cout_g, cin, 3, 3)).astype(np.float32)
        wp = H.pack_weight_stack([torch.from_numpy(w[g]).cuda() for g in range(G)])
        frag = H.winograd_weight(wp).cpu().numpy()
        ck, mb = lib.vsp_conv2d_winograd_chunk(), lib.vsp_conv2d_winograd_mbw(cout_g)
        nch, ntile = (cin + ck - 1) // ck, (cout_g + 16 * mb - 1) // (16 * mb)
        assert frag.size == G * ntile * nch * 8 * 64 * 2 * mb

        def U(g, pos, ci, co):  # the index formula of include/vspbfr_hip.h
            wave, pp = pos // 2, pos % 2
            chunk, kq = ci // ck, ci % ck
            tile, rem = co // (16 * mb), co % (16 * mb)
            m, lr = rem // 16, rem % 16
            lane = kq * 16 + lr
            return frag[((((((g * ntile + tile) * nch + chunk) * 8 + wave) * 2 + pp) * 64 + lane) * mb + m)]

        Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
        At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
        x = rng.standard_normal((cin, 8, 8))
        for g in range(G):
            d = dils[g]
            ref = F.conv2d(torch.from_numpy(x)[None].float(), torch.from_numpy(w[g]), padding=d, dilation=d)[0].numpy()
            Ug = np.array([[[U(g, pos, ci, co) for co in range(cout_g)] for ci in range(cin)] for pos in range(16)])
            xp = np.pad(x, ((0, 0), (d, d), (d, d)))
            out = np.zeros((cout_g, 8, 8))
            for ry in range(d):           # polyphase: residue (ry, rx) of the dilation-d grid is a dilation-1 problem
                for rx in range(d):
                    sub = xp[:, ry::d, rx::d]                      # padded sub-image (pad 1 in sub coordinates)
                    sh, sw = (8 - ry + d - 1) // d, (8 - rx + d - 1) // d
                    for ty in range(0, sh, 2):
                        for tx in range(0, sw, 2):
                            win = np.zeros((cin, 4, 4))
                            blk = sub[:, ty:ty + 4, tx:tx + 4]
                            win[:, :blk.shape[1], :blk.shape[2]] = blk
                            V = np.einsum("ar,crs,bs->cab", Bt, win, Bt).reshape(cin, 16)
                            M = np.einsum("pio,ip->op", Ug, V).reshape(cout_g, 4, 4)
                            Y = np.einsum("ia,oab,jb->oij", At, M, At)
                            for i in range(2):
                                for j in range(2):
                                    oy, ox = (ty + i) * d + ry, (tx + j) * d + rx
                                    if ty + i < sh and tx + j < sw and oy < 8 and ox < 8:
                                        out[:, oy, ox] = Y[:, i, j]
            assert np.abs(out - ref).max() < 2e-4, (G, cin, cout_g, g, np.abs(out - ref).max())


def test_bf16_weight_lds_image_layout_cpu():
    """hip_ops.bf16_weight against the documented index formula of vsp_conv2d_bf16 (include/vspbfr_hip.h): reading the packed
    bf16 words back through the formula and convolving on the host reproduces F.conv2d on the bf16-rounded weights; and the
    eligibility rules of the bf16 configuration (no GPU needed)."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    from vspbfr_amd import hip_ops as H
    rng = np.random.default_rng(7)
    for G, cin, cout_g in ((1, 24, 20), (4, 16, 16), (2, 40, 70)):
        w = rng.standard_normal((G, cout_g, cin, 3, 3)).astype(np.float32)
        wp = torch.stack([H.pack_weight(torch.from_numpy(w[g]))[0] for g in range(G)])
        packed = H.bf16_weight(wp)
        assert packed.dtype == torch.bfloat16
        nch, co_pad = (cin + 15) // 16, (cout_g + 31) // 32 * 32
        assert packed.numel() == G * nch * 9 * 2 * co_pad * 8
        flat = packed.float().numpy()

        def W(g, tap, ci, co):  # the index formula of include/vspbfr_hip.h
            chunk, octet, j = ci // 16, (ci % 16) // 8, ci % 8
            return flat[(((((g * nch + chunk) * 9 + tap) * 2 + octet) * co_pad + co) * 8) + j]

        for g in range(G):
            back = np.array([[[[W(g, ky * 3 + kx, ci, co) for kx in range(3)] for ky in range(3)] for ci in range(cin)]
                             for co in range(cout_g)], np.float32)
            assert np.array_equal(back, torch.from_numpy(w[g]).to(torch.bfloat16).float().numpy())
        # zero padding of the channel octets past Cin and of the output-channel rows past cout_g
        full = flat.reshape(G, nch, 9, 2, co_pad, 8)
        assert not full[:, :, :, :, cout_g:, :].any()
        if cin % 16:
            assert not full[:, -1, :, 1 if cin % 16 <= 8 else 2:, :, :].any()
    # split-precision packing (vsp_conv2d_bf16x3): part 0 + part 1 reproduce the weight to 2^-16, parts are chunk-interleaved
    w = rng.standard_normal((2, 20, 24, 3, 3)).astype(np.float32)
    wp = torch.stack([H.pack_weight(torch.from_numpy(w[g]))[0] for g in range(2)])
    both = H.bf16x3_weight(wp).float().numpy().reshape(2, 2, 2, 9, 2, 32, 8)   # [g][chunk][part][tap][octet][co_pad][8]
    rec = both[:, :, 0] + both[:, :, 1]                                        # [g][chunk][tap][octet][co][8]
    for g in range(2):
        for ci in range(24):
            got = rec[g, ci // 16, :, (ci % 16) // 8, :20, ci % 8]                    # [tap][co]
            want = w[g, :, ci].reshape(20, 9).T
            assert np.abs(got - want).max() <= np.abs(want).max() * 2.0 ** -15
    x = torch.from_numpy(rng.standard_normal((1, 24, 9, 9)).astype(np.float32))
    w0 = torch.from_numpy(rng.standard_normal((20, 24, 3, 3)).astype(np.float32))
    ref = F.conv2d(x.to(torch.bfloat16).float(), w0.to(torch.bfloat16).float(), padding=1)
    assert torch.isfinite(ref).all()
    pc = lambda **kw: H.PackedConv(torch.zeros(1), kw.get("G", 1), 32, kw.get("cin", 64), kw.get("k", 3), kw.get("k", 3),
                                   kw.get("stride", 1), kw.get("dil", (1,)), kw.get("pad", (1,)), x_group_stride=kw.get("xgs", 0))
    assert H.bf16_eligible(pc(), 64, 64, 64, 64)
    assert H.bf16_eligible(pc(G=4, dil=(1, 2, 4, 8), pad=(1, 2, 4, 8)), 64, 64, 64, 64)
    assert H.bf16_eligible(pc(stride=2, pad=(0,)), 65, 65, 32, 32) and H.bf16_eligible(pc(stride=2), 64, 64, 32, 32)
    assert H.bf16_eligible(pc(), 32, 32, 33, 33, transposed=True)
    assert not H.bf16_eligible(pc(k=1, pad=(0,)), 64, 64, 64, 64)          # 1x1
    assert not H.bf16_eligible(pc(cin=12), 64, 64, 64, 64)                 # Cin not a multiple of 8
    assert not H.bf16_eligible(pc(stride=2, dil=(2,), pad=(2,)), 64, 64, 32, 32)
    assert not H.bf16_eligible(pc(G=8), 64, 64, 64, 64)                    # > 4 groups over one shared input
    assert H.bf16_eligible(pc(G=8, xgs=64), 64, 64, 64, 64)                # true groups
    assert not H.bf16_profitable(pc(), 8, 8, 8, 8) and H.bf16_profitable(pc(), 16, 16, 16, 16)


def test_bf16rv_weight_fragment_layout_cpu():
    """hip_ops.bf16rv_weight against the documented index formula of vsp_conv2d_bf16rv (include/vspbfr_hip.h): every (tap, ci, co) is
    found where the formula says, the fourth horizontal slot and the channel rows past cout_g are zero; eligibility and the choice rule of
    the row-vector kernel (no GPU needed)."""
    import numpy as np
    import torch
    from vspbfr_amd import hip_ops as H
    rng = np.random.default_rng(11)
    for G, cin, cout_g in ((1, 24, 64), (4, 16, 16), (2, 8, 40)):
        w = rng.standard_normal((G, cout_g, cin, 3, 3)).astype(np.float32)
        wp = torch.stack([H.pack_weight(torch.from_numpy(w[g]))[0] for g in range(G)])
        packed = H.bf16rv_weight(wp)
        co_pad, nch = (cout_g + 31) // 32 * 32, cin // 8
        assert packed.dtype == torch.bfloat16 and packed.numel() == G * nch * 3 * 2 * 2 * co_pad * 8
        flat = packed.float().numpy()

        def W(g, ky, kx, ci, co):   # the header's formula, per group
            chunk, quad, half, pair = ci // 8, (ci % 8) // 4, (ci % 4) // 2, ci % 2
            return flat[g * (nch * 12 * co_pad * 8) + (((((chunk * 3 + ky) * 2 + quad) * 2 + half) * co_pad + co) * 8) + pair * 4 + kx]

        for g in range(G):
            back = np.array([[[[W(g, ky, kx, ci, co) for kx in range(3)] for ky in range(3)] for ci in range(cin)] for co in range(cout_g)], np.float32)
            assert np.array_equal(back, torch.from_numpy(w[g]).to(torch.bfloat16).float().numpy())
        full = flat.reshape(G, nch, 3, 2, 2, co_pad, 2, 4)
        assert not full[..., 3].any() and not full[:, :, :, :, :, cout_g:].any()
    pc = lambda **kw: H.PackedConv(torch.zeros(1), kw.get("G", 1), kw.get("cout_g", 64), kw.get("cin", 64), kw.get("k", 3), kw.get("k", 3),
                                   kw.get("stride", 1), kw.get("dil", (1,)), kw.get("pad", (1,)), x_group_stride=kw.get("xgs", 0))
    assert H.bf16rv_eligible(pc(), 128, 128, 128, 128) and H.bf16rv_eligible(pc(cin=32, cout_g=32), 1024, 1024, 1024, 1024)
    assert H.bf16rv_eligible(pc(G=4, cout_g=16, dil=(1, 2, 4, 8), pad=(1, 2, 4, 8)), 512, 512, 512, 512)
    assert not H.bf16rv_eligible(pc(), 128, 96, 128, 96)                                   # W % 64
    assert not H.bf16rv_eligible(pc(cin=512), 64, 64, 64, 64)                              # more than 256 input channels
    assert not H.bf16rv_eligible(pc(cout_g=48), 128, 128, 128, 128)                        # plain layers: channels % 32
    assert not H.bf16rv_eligible(pc(stride=2), 128, 128, 64, 64) and not H.bf16rv_eligible(pc(dil=(2,), pad=(2,)), 128, 128, 128, 128)
    assert not H.bf16rv_eligible(pc(G=4, cout_g=16, dil=(1, 2, 3, 8), pad=(1, 2, 3, 8)), 512, 512, 512, 512)   # dilations from {1, 2, 4, 8}
    assert H.bf16rv_profitable(pc(), 128, 128) and not H.bf16rv_profitable(pc(), 64, 64)
    assert not H.bf16rv_profitable(pc(G=4, cout_g=16, dil=(1, 2, 4, 8), pad=(1, 2, 4, 8)), 512, 512)    # served, not chosen


def test_tacc_projection_fragment_layout():
    """vsp_tacc_block.wcat_frag (include/vspbfr_hip.h): the concatenated [4D, D] projection matrix in MFMA fragment order --
    frag[n / 16][k / 16][lane = 16 (k % 16 / 4) + n % 16][k % 4] = W[n][k] -- as diffusion.Code_diffuser builds it (host logic, CPU)."""
    import torch
    from vspbfr_amd.diffusion import Code_diffuser
    net = Code_diffuser(timesteps=4)
    blk = net.att_mapper[0]
    W = net._wcat(blk)
    F_ = net._wcat_frag(blk)
    n_, k_ = W.shape
    assert (n_, k_) == (2048, 512) and F_.numel() == W.numel() and F_.is_contiguous()
    flat = F_.reshape(n_ // 16, k_ // 16, 64, 4)
    g = torch.Generator().manual_seed(0)
    for n, k in zip(torch.randint(0, n_, (200,), generator=g).tolist(), torch.randint(0, k_, (200,), generator=g).tolist()):
        lane = 16 * ((k % 16) // 4) + n % 16
        assert flat[n // 16, k // 16, lane, k % 4] == W[n, k]
    # the rows are the four projections in the order the kernels slice them: k, v, q2, v2
    srcs = [blk.k_matrix.weight, blk.v_matrix.weight, blk.attention_layer.q_matrix.weight, blk.attention_layer.v_matrix.weight]
    assert torch.equal(W, torch.cat([w.detach() for w in srcs], 0))

