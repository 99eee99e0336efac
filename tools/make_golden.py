"""Generate tests/golden/* by running the REAL reference (/root/reference) on CPU.  BUILD CONTAINER ONLY.

    python tools/make_golden.py [--only ops,layers,diffuser,restorenet64,generator64,encoder,pipeline512]

Inputs and weights come from oracle.keyed_rng / oracle.weights (regenerable anywhere from names), so the fixtures hold
only the reference's OUTPUTS (plus the state-dict key/shape lists of its modules).  Every random draw of the reference
is replaced by an explicit tensor: NoiseInjection.forward is monkeypatched in-process (no reference file is edited) to
consume a queue, because Restoration_net's own `noise=` argument is unusable.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import refshim  # noqa: E402

refshim.install()

from oracle import cases, weights  # noqa: E402
from oracle import models as omodels  # noqa: E402

import models.RestoreNet as RN  # noqa: E402  (reference)
from models.CodeDiffuser import Code_diffuser  # noqa: E402
from ldm.ddpm import My_DDPM  # noqa: E402
import e4e.models.stylegan2.model as SG  # noqa: E402
from e4e.models.encoders.psp_encoders import Encoder4Editing  # noqa: E402
from op import fused_leaky_relu as ref_fused_leaky_relu  # noqa: E402
sys_upfirdn = sys.modules["op.upfirdn2d"]

GOLD = os.path.join(ROOT, "tests", "golden")
torch.set_grad_enabled(False)

# ---- explicit-noise hook -------------------------------------------------------------------------------------------
NOISE_QUEUE = []


def _patched_noise_forward(self, image, noise=None):
    if noise is None:
        noise = NOISE_QUEUE.pop(0)
        assert noise.shape == (image.shape[0], 1, image.shape[2], image.shape[3]), (noise.shape, image.shape)
    return image + self.weight * noise


RN.NoiseInjection.forward = _patched_noise_forward
SG.NoiseInjection.forward = _patched_noise_forward


def spec_of(module):
    return [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in module.state_dict().items()]


def load_synth(module, model_kind, seed):
    sd = weights.synth_state_dict(model_kind, spec_of(module), seed)
    module.load_state_dict(sd, strict=True)
    module.eval()
    return sd


def np_(t):
    return t.detach().cpu().numpy()


# ---- generators ----------------------------------------------------------------------------------------------------
def gen_specs():
    from argparse import Namespace
    specs = {
        "restorenet512": spec_of(RN.Restoration_net(512, 512, 8)),
        "restorenet64": spec_of(RN.Restoration_net(64, 512, 8)),
        "diffuser": spec_of(Code_diffuser(timesteps=4)),
        "e4e_encoder": spec_of(Encoder4Editing(50, "ir_se", Namespace(input_channel=3, stylegan_size=1024))),
        "e4e_decoder1024": spec_of(SG.Generator(1024, 512, 8, channel_multiplier=2)),
        "e4e_decoder64": spec_of(SG.Generator(64, 512, 8, channel_multiplier=2)),
        "discriminator512": spec_of(RN.Discriminator(512)),
        "discriminator64": spec_of(RN.Discriminator(64)),
    }
    path = os.path.join(GOLD, "state_specs.json")
    if os.path.exists(path):   # entries other generators add (gen_lpips, gen_idloss) survive a re-run of this one
        with open(path) as f:
            specs = {**{k: v for k, v in json.load(f).items() if k in ("lpips_vgg", "arcface_resnet101")}, **specs}
    with open(path, "w") as f:
        json.dump(specs, f)
    print("state_specs.json:", {k: len(v) for k, v in specs.items()})


def gen_ops():
    out = {}
    for name in cases.LRELU_CASES:
        x, b = cases.lrelu_inputs(name)
        out[name] = np_(ref_fused_leaky_relu(x, b))
    for name in cases.FIR_CASES:
        x, k, up, down, pad = cases.fir_inputs(name)
        out[name] = np_(sys_upfirdn.upfirdn2d_native(x, k, up[0], up[1], down[0], down[1], *pad))
    np.savez_compressed(os.path.join(GOLD, "ops.npz"), **out)
    print("ops.npz:", {k: v.shape for k, v in out.items()})


def gen_layers():
    out = {}
    for name, (kind, cin, cout, k, sdim, xs, extra) in cases.MODCONV_CASES.items():
        m = RN.ModulatedConv2d(cin, cout, k, sdim, upsample=(kind == "up"), downsample=(kind == "down"), **extra)
        named = [(n, tuple(v.shape)) for n, v in m.state_dict().items()]
        m.load_state_dict(cases.module_weights(name, named))
        x, style = cases.tensor(name, "x", xs), cases.tensor(name, "style", (xs[0], sdim))
        out[name] = np_(m(x, style))
    for name, (cin, cout, xs, d) in cases.DILCONV_CASES.items():
        m = RN.Dilated_ModulatedConv2d(cin, cout, 3, 16, dilation=d)
        named = [(n, tuple(v.shape)) for n, v in m.state_dict().items()]
        m.load_state_dict(cases.module_weights(name, named))
        x, style = cases.tensor(name, "x", xs), cases.tensor(name, "style", (xs[0], cin)) * 0.5 + 1.0
        out[name] = np_(m(x, style))
    for name, (cin, cout, sdim, xs) in cases.SMART_CASES.items():
        m = RN.SMART_layer(cin, cout, 3, sdim)
        named = [(n, tuple(v.shape)) for n, v in m.state_dict().items()]
        m.load_state_dict(cases.module_weights(name, named))
        x, style = cases.tensor(name, "x", xs), cases.tensor(name, "style", (xs[0], sdim))
        noise = cases.tensor(name, "noise", (xs[0], 1, xs[2], xs[3]))
        out[name] = np_(m(x, style, noise=noise))
    for name, (cin, cout, k, xs) in cases.LARGECONV_CASES.items():
        m = RN.LargeConvLayer(cin, cout, kernel_size=k)
        named = [(n, tuple(v.shape)) for n, v in m.state_dict().items()]
        m.load_state_dict(cases.module_weights(name, named))
        out[name] = np_(m(cases.tensor(name, "x", xs)))
    np.savez_compressed(os.path.join(GOLD, "layers.npz"), **out)
    print("layers.npz:", {k: v.shape for k, v in out.items()})


def gen_diffuser():
    out = {}
    for name, (B, T, ls, le) in cases.DIFFUSER_CASES.items():
        net = Code_diffuser(timesteps=T)
        load_synth(net, "diffuser", cases.SEED)
        ddpm = My_DDPM(denoise=net, linear_start=ls, linear_end=le, timesteps=T)
        cond, x_T = cases.diffuser_inputs(name)
        # one denoiser call (unit check) + the whole sampling loop restated with the public p_sample and our x_T
        # (My_DDPM.forward draws x_T itself, ldm/ddpm.py:423; the loop body is ldm/ddpm.py:426-428)
        t = torch.full((B,), T - 1, dtype=torch.long)
        out[name + "/x0_first"] = np_(net(x_T, cond, t))
        x = x_T
        for i in reversed(range(T)):
            x, _ = ddpm.p_sample(x, torch.full((B,), i, dtype=torch.long), cond, clip_denoised=ddpm.clip_denoised)
        out[name + "/final"] = np_(x)
        out[name + "/coef1"] = np_(ddpm.posterior_mean_coef1)
        out[name + "/coef2"] = np_(ddpm.posterior_mean_coef2)
    np.savez_compressed(os.path.join(GOLD, "diffuser.npz"), **out)
    print("diffuser.npz:", {k: v.shape for k, v in out.items()})


def gen_ddim():
    """BASELINE config 3: DDIMSampler (dead code in the reference, ldm/ddim.py) driven the only way it can run on this
    path: flatten adapter for its (b, L) latents and model.model(x, t, c) argument order,
    register_buffer patched to plain setattr (it hard-moves buffers to "cuda"), S < T."""
    from ldm.ddim import DDIMSampler
    B, T, S = 2, 50, 25
    net = Code_diffuser(timesteps=T)
    load_synth(net, "diffuser", cases.SEED)
    ddpm = My_DDPM(denoise=net, timesteps=T)  # default betas (1e-4, 2e-2)
    cond, x_T = cases.diffuser_inputs("ddim_T50_S25")

    class Adapter(torch.nn.Module):
        def forward(self, x, t, c):
            return net(x.view(B, 18, 512), c, t).reshape(B, -1)

    ddpm.model = Adapter()
    DDIMSampler.register_buffer = lambda self, name, attr: setattr(self, name, attr)
    sampler = DDIMSampler(ddpm, device="cpu")
    samples, _ = sampler.sample(S=S, batch_size=B, shape=18 * 512, conditioning=cond, eta=0.0, verbose=False,
                                x_T=x_T.reshape(B, -1))
    out = {"final": np_(samples.view(B, 18, 512)), "ddim_timesteps": np.asarray(sampler.ddim_timesteps),
           "ddim_alphas": np.asarray(sampler.ddim_alphas), "ddim_alphas_prev": np.asarray(sampler.ddim_alphas_prev)}
    np.savez_compressed(os.path.join(GOLD, "ddim.npz"), **out)
    print("ddim.npz:", out["final"].shape, "absmax %.3f" % np.abs(out["final"]).max(), out["ddim_timesteps"][:4], "...")


def run_restorenet(size, B, case, net=None, de_feats=None, pre_styles=None):
    net = net or RN.Restoration_net(size, 512, 8)
    load_synth(net, "restorenet", cases.SEED)
    imgs = cases.image_batch(case, B, size)
    enc_s, dec_s = omodels.restoration_noise_shapes(size, B)
    enc_noise, dec_noise = cases.noise_list(case, "enc", enc_s), cases.noise_list(case, "dec", dec_s)
    log_size = int(np.log2(size))
    if de_feats is None:
        chans = net.channels
        de_feats = [cases.tensor(case, f"de_feat{k}", (B, chans[2 ** (k + 2)], 2 ** (k + 2), 2 ** (k + 2)), 0.5)
                    for k in range(log_size - 1)]
        pre_styles = cases.tensor(case, "pre_styles", (B, 18, 512))
    z = cases.tensor(case, "z", (B, 512))
    NOISE_QUEUE.clear()
    NOISE_QUEUE.extend(enc_noise + dec_noise)
    img = net(imgs, de_feats, pre_styles, [z])
    assert not NOISE_QUEUE
    return img


def gen_restorenet64():
    t = time.time()
    img = run_restorenet(64, 1, "restorenet64")
    np.savez_compressed(os.path.join(GOLD, "restorenet64.npz"), image=np_(img))
    print("restorenet64.npz:", tuple(img.shape), "std %.3f absmax %.3f" % (img.std(), img.abs().max()), "%.1fs" % (time.time() - t))


GRAD_SAMPLES = 2048


def grad_sample(g):
    """A strided sample of a gradient tensor (fixtures stay small) -- the test takes the same sample of its own gradient."""
    f = g.detach().reshape(-1)
    return f[::max(1, f.numel() // GRAD_SAMPLES)][:GRAD_SAMPLES]


def gen_restorenet64_grad():
    """The generator half of the training step (restoration_train.py:153-255) at size 64: forward + backward of the REFERENCE
    Restoration_net in eval mode (Dropout2d off, so the pass is deterministic) for the scalar loss <image, R>, R a keyed random
    tensor: the image, the loss and a strided sample of the gradient of every parameter and of the differentiable inputs
    (pre_styles, the prior's features).  B = 2 so that the groups = batch convolutions really carry two groups."""
    t = time.time()
    case, size, B = "restorenet64_grad", 64, 2
    net = RN.Restoration_net(size, 512, 8)
    load_synth(net, "restorenet", cases.SEED)
    imgs = cases.image_batch(case, B, size)
    enc_s, dec_s = omodels.restoration_noise_shapes(size, B)
    enc_noise, dec_noise = cases.noise_list(case, "enc", enc_s), cases.noise_list(case, "dec", dec_s)
    de_feats = [cases.tensor(case, f"de_feat{k}", (B, net.channels[2 ** (k + 2)], 2 ** (k + 2), 2 ** (k + 2)), 0.5).requires_grad_(True)
                for k in range(5)]
    pre = cases.tensor(case, "pre_styles", (B, 18, 512)).requires_grad_(True)
    z = cases.tensor(case, "z", (B, 512))
    R = cases.tensor(case, "R", (B, 3, size, size))
    NOISE_QUEUE.clear()
    NOISE_QUEUE.extend(enc_noise + dec_noise)
    with torch.enable_grad():
        img = net(imgs, de_feats, pre, [z])
        loss = (img * R).sum()
        loss.backward()
    assert not NOISE_QUEUE
    out = {"image": np_(img), "loss": np.array([loss.item()], dtype=np.float64), "d_pre_styles": np_(grad_sample(pre.grad)),
           "d_pre_styles_norm": np.array([pre.grad.norm().item()])}
    for k, f in enumerate(de_feats):   # (de_feats[0] never enters the decoder: models/RestoreNet.py:1030-1035 start at k = 1)
        if f.grad is not None:
            out[f"d_de_feat{k}"] = np_(grad_sample(f.grad))
    names = []
    for name, p_ in net.named_parameters():
        if p_.grad is None:
            continue
        names.append(name)
        out["g/" + name] = np_(grad_sample(p_.grad))
        out["n/" + name] = np.array([p_.grad.norm().item()])
    out["param_names"] = np.array(names)
    np.savez_compressed(os.path.join(GOLD, "restorenet64_grad.npz"), **out)
    print("restorenet64_grad.npz: loss %.4f, %d parameter gradients, |d pre| %.3e, %.1fs" % (loss.item(), len(names), pre.grad.norm().item(), time.time() - t))


def gen_discriminator64():
    """The discriminator half of the training step (restoration_train.py:176-218, 60-79) at size 64, batch 4 (= the minibatch
    stddev group): the REFERENCE Discriminator's predictions on a "real" and a "fake" keyed batch, the logistic loss with its
    parameter gradients, the R1 penalty (a double backward through every layer) with its parameter gradients, and the
    generator's non-saturating loss with its gradient w.r.t. the fake image."""
    import torch.nn.functional as F
    from torch import autograd
    import op.conv2d_gradfix as ref_gradfix
    t = time.time()
    case, size, B = "discriminator64", 64, 4
    D = RN.Discriminator(size)
    load_synth(D, "discriminator", cases.SEED)
    real, fake = cases.image_batch(case + "/real", B, size), cases.image_batch(case + "/fake", B, size)
    out = {}
    with torch.enable_grad():
        D.zero_grad()
        rp, fp = D(real), D(fake)
        d_loss = F.softplus(-rp).mean() + F.softplus(fp).mean()
        d_loss.backward()
        out.update(real_pred=np_(rp), fake_pred=np_(fp), d_loss=np.array([d_loss.item()]))
        names = [n for n, p_ in D.named_parameters() if p_.grad is not None]
        for n, p_ in D.named_parameters():
            out["gd/" + n], out["nd/" + n] = np_(grad_sample(p_.grad)), np.array([p_.grad.norm().item()])
        D.zero_grad()
        x = real.detach().clone().requires_grad_(True)
        pred = D(x)
        with ref_gradfix.no_weight_gradients():
            grad_real, = autograd.grad(outputs=pred.sum(), inputs=x, create_graph=True)
        r1 = grad_real.pow(2).reshape(B, -1).sum(1).mean()
        (10.0 / 2 * r1 * 16 + 0 * pred[0]).backward()          # restoration_train.py:211 with r1 = 10, d_reg_every = 16
        out.update(r1=np.array([r1.item()]), grad_real=np_(grad_sample(grad_real)))
        for n, p_ in D.named_parameters():
            out["gr/" + n], out["nr/" + n] = np_(grad_sample(p_.grad)), np.array([p_.grad.norm().item()])
        D.zero_grad()
        xf = fake.detach().clone().requires_grad_(True)
        g_loss = F.softplus(-D(xf)).mean()
        g_loss.backward()
        out.update(g_loss=np.array([g_loss.item()]), d_fake_image=np_(grad_sample(xf.grad)), d_fake_image_norm=np.array([xf.grad.norm().item()]))
    out["param_names"] = np.array(names)
    np.savez_compressed(os.path.join(GOLD, "discriminator64.npz"), **out)
    print("discriminator64.npz: d_loss %.4f r1 %.4e g_loss %.4f, %d parameters, %.1fs" % (d_loss.item(), r1.item(), g_loss.item(), len(names), time.time() - t))


def gen_ada_draws():
    """Seeded draws of the REFERENCE's sample_affine / sample_color (non_leaking.py:660-760): the host-side matrices of ADA, including the
    quarter-turn categories (0, 3) of :673.  vspbfr_amd.non_leaking follows the same torch RNG call sequence, so the same seed must
    give the same matrices (tests/test_ada_draws.py, CPU)."""
    import non_leaking as NL
    out = {}
    for seed, p_aug, size in ((5, 0.8, 64), (23, 0.35, 256)):
        torch.manual_seed(seed)
        out[f"G_{seed}"] = np_(NL.sample_affine(p_aug, 32, size, size))
        out[f"C_{seed}"] = np_(NL.sample_color(p_aug, 32))
    out["cases"] = np.array([[5, 0.8, 64], [23, 0.35, 256]])
    np.savez_compressed(os.path.join(GOLD, "ada_draws.npz"), **out)
    print("ada_draws.npz:", {k: v.shape for k, v in out.items()})


def gen_ada():
    """ADA augmentation (non_leaking.py:857-934) of the REFERENCE with pinned transformation matrices: G = the inverse of a
    sample_affine draw, C = a sample_color draw (both stored), the augmented batch, and the gradient of <augmented, R> w.r.t.
    the input image (the path the generator's gradient takes when augmentation is on)."""
    import non_leaking as NL
    import torch.nn.functional as F
    # GridSampleBackward (non_leaking.py:826-846) fetches aten::grid_sampler_2d_backward through torch._C._jit_get_operation, which
    # returns a tuple on this torch (2.10): the reference's own backward cannot run.  Its forward is exactly this call (:811-813);
    # in-process replacement by the plain op, whose backward autograd knows (no reference file is edited)
    NL.grid_sample = lambda inp, grid: F.grid_sample(inp, grid, mode="bilinear", padding_mode="zeros", align_corners=False)
    case, B, size, p_aug = "ada", 4, 64, 0.8
    img = cases.image_batch(case, B, size).requires_grad_(True)
    R = cases.tensor(case, "R", (B, 3, size, size))
    torch.manual_seed(11)
    G = torch.inverse(NL.sample_affine(p_aug, B, size, size))
    C = NL.sample_color(p_aug, B)
    with torch.enable_grad():
        out, _ = NL.augment(img, p_aug, (G, C))
        (out * R).sum().backward()
    geo, _ = NL.random_apply_affine(img.detach(), p_aug, G)
    np.savez_compressed(os.path.join(GOLD, "ada.npz"), G=np_(G), C=np_(C), out=np_(out), geometric=np_(geo), d_img=np_(img.grad))
    print("ada.npz:", tuple(out.shape), "std %.3f" % out.std(), "G[0]", G[0].tolist())


def gen_generator64():
    g = SG.Generator(64, 512, 8, channel_multiplier=2)
    load_synth(g, "e4e_decoder", cases.SEED)
    B = 2
    latent = cases.tensor("generator64", "latent", (B, 10, 512))
    noise = cases.noise_list("generator64", "n", omodels.generator_noise_shapes(64, B))
    img, feats = g([latent], input_is_latent=True, noise=noise, return_features=True)
    out = {"image": np_(img)}
    for i, f in enumerate(feats):
        out[f"feat{i}"] = np_(cases.feat_sample(f))
    np.savez_compressed(os.path.join(GOLD, "generator64.npz"), **out)
    print("generator64.npz:", {k: v.shape for k, v in out.items()}, "img std %.3f" % img.std())


def gen_encoder():
    from argparse import Namespace
    enc = Encoder4Editing(50, "ir_se", Namespace(input_channel=3, stylegan_size=1024))
    load_synth(enc, "e4e_encoder", cases.SEED)
    x = cases.image_batch("encoder", 1, 256)
    w = enc(x)
    np.savez_compressed(os.path.join(GOLD, "encoder.npz"), codes=np_(w))
    print("encoder.npz:", tuple(w.shape), "std %.3f absmax %.3f" % (w.std(), w.abs().max()))


def gen_pipeline512():
    """A+B+C+D at 512^2, B=1, T=4 (restoration_test.py:125-131) with every draw pinned."""
    from argparse import Namespace
    t0 = time.time()
    case = "pipeline512"
    B, T = 1, 4
    enc = Encoder4Editing(50, "ir_se", Namespace(input_channel=3, stylegan_size=1024))
    load_synth(enc, "e4e_encoder", cases.SEED)
    dec = SG.Generator(1024, 512, 8, channel_multiplier=2)
    load_synth(dec, "e4e_decoder", cases.SEED)
    latent_avg = weights.synth_tensor("e4e_decoder", "latent_avg", (18, 512), "float32", cases.SEED)
    net = Code_diffuser(timesteps=T)
    load_synth(net, "diffuser", cases.SEED)
    ddpm = My_DDPM(denoise=net, linear_start=0.1, linear_end=0.99, timesteps=T)
    lq = cases.image_batch(case, B, 512)
    # A: E4e_embedding.get_w_plus (Loss/e4e_embedding.py:91-100) + My_pSp.forward (e4e/models/psp.py:145-165)
    x256 = torch.nn.functional.interpolate(lq, (256, 256), mode="bilinear")
    codes = (enc(x256) + latent_avg.repeat(B, 1, 1))[:, :18]
    # B: My_DDPM.forward(training=False) with x_T supplied
    x = cases.tensor(case, "x_T", (B, 18, 512))
    for i in reversed(range(T)):
        x, _ = ddpm.p_sample(x, torch.full((B,), i, dtype=torch.long), codes, clip_denoised=ddpm.clip_denoised)
    pre = x
    # C: My_pSp.stylegan2_feat_forward (e4e/models/psp.py:235-248)
    gnoise = cases.noise_list(case, "g", omodels.generator_noise_shapes(1024, B))
    img1024, feats = dec([pre], input_is_latent=True, noise=gnoise, return_features=True)
    feats = feats[:16]
    sample = torch.nn.AdaptiveAvgPool2d((512, 512))(img1024)
    # D
    restored = run_restorenet(512, B, case, de_feats=feats, pre_styles=pre)
    out = {
        "codes": np_(codes), "pre_latent": np_(pre),
        "sample_sub": np_(sample[:, :, ::8, ::8]), "restored_sub": np_(restored[:, :, ::8, ::8]),
        "restored_crop": np_(restored[:, :, 200:264, 200:264]),
        "restored_stats": np.array([restored.mean(), restored.std(), restored.abs().max()], dtype=np.float32),
        "sample_stats": np.array([sample.mean(), sample.std(), sample.abs().max()], dtype=np.float32),
    }
    for i, f in enumerate(feats):
        out[f"feat{i}_stats"] = np.array([f.mean(), f.std(), f.abs().max()], dtype=np.float32)
    np.savez_compressed(os.path.join(GOLD, "pipeline512.npz"), **out)
    print("pipeline512.npz: restored stats", out["restored_stats"], "sample stats", out["sample_stats"], "%.1fs" % (time.time() - t0))


def gen_loader():
    """The test-time loaders of the reference (dataset.py:376-495: ImageFolder_restore_test / _no_gt) on a small committed
    image folder (tests/golden/loader_images: written here, deterministic): sorted recursive listing, extension filter,
    LANCZOS resize-to-cover + centre crop.  `transform=None`: the classes return the PIL image; the caller's transform is
    ToTensor + Normalize(0.5, 0.5) (restoration_test.py:89-94, torchvision, not installed here) which the test applies.
    dataset.py imports my_basicsr.my_degradations -> torchvision.transforms.functional.rgb_to_grayscale at module level
    (training-time degradations, never reached by these classes): an empty stand-in module lets the import through."""
    import types
    from PIL import Image
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision.transforms.functional"].rgb_to_grayscale = None
    import dataset as ref_dataset
    root = os.path.join(GOLD, "loader_images")
    os.makedirs(os.path.join(root, "lq", "sub"), exist_ok=True)
    os.makedirs(os.path.join(root, "hq"), exist_ok=True)
    yy, xx = np.mgrid[0:1:200j, 0:1:200j]

    def picture(w, h, k):  # smooth, compressible, different per k
        y, x = yy[:h, :w] * (1 + 0.3 * k), xx[:h, :w] * (1 + 0.2 * k)
        rgb = np.stack([np.sin(6 * x + k) * np.cos(4 * y), np.cos(5 * x * y + 0.5 * k), np.sin(3 * (x - y) + k)], -1)
        return Image.fromarray(np.clip((rgb * 0.5 + 0.5) * 255, 0, 255).astype(np.uint8))
    files = [("lq/b_wide.png", 100, 70), ("lq/a_exact.png", 64, 64), ("lq/sub/tall_small.png", 30, 50), ("lq/c_photo.jpg", 90, 81),
             ("lq/notes.txt", 0, 0), ("hq/a_exact.png", 64, 64), ("hq/b_wide.png", 120, 80), ("hq/c_photo.png", 90, 81), ("hq/d_tall.png", 60, 100)]
    for k, (rel, w, h) in enumerate(files):
        path = os.path.join(root, rel)
        if rel.endswith(".txt"):
            open(path, "w").write("not an image\n")
        else:
            picture(w, h, k).save(path, quality=90) if rel.endswith(".jpg") else picture(w, h, k).save(path)
    out = {}
    ds = ref_dataset.ImageFolder_restore_test_no_gt(lq_root=os.path.join(root, "lq"), transform=None, im_size=(64, 64))
    out["no_gt/files"] = np.array([os.path.relpath(f, root) for f in ds.lq_frame])
    for i in range(len(ds)):
        out[f"no_gt/{i}"] = np.asarray(ds[i], dtype=np.uint8)
    ds2 = ref_dataset.ImageFolder_restore_test(lq_root=os.path.join(root, "lq"), hq_root=os.path.join(root, "hq"), transform=None,
                                               im_size=(64, 64))
    out["gt/lq_files"] = np.array([os.path.relpath(f, root) for f in ds2.lq_frame])
    out["gt/hq_files"] = np.array([os.path.relpath(f, root) for f in ds2.hq_frame])
    for i in range(len(ds2)):
        lq, hq = ds2[i]
        out[f"gt/{i}/lq"], out[f"gt/{i}/hq"] = np.asarray(lq, dtype=np.uint8), np.asarray(hq, dtype=np.uint8)
    np.savez_compressed(os.path.join(GOLD, "loader.npz"), **out)
    print("loader.npz:", list(out["no_gt/files"]), {k: v.shape for k, v in out.items() if not k.endswith("files")})


def gen_lpips():
    """The perceptual term of the generator loss (restoration_train.py:143, 236-239): the REFERENCE's
    my_lpips.PerceptualLoss(model="net-lin", net="vgg") -- ScalingLayer, five VGG16 slices, normalize_tensor, squared difference,
    the `lin` layers with the reference's OWN v0.1 weights (my_lpips/weights/v0.1/vgg.pth, stored in the fixture), spatial
    average -- on keyed 64x64 batches with a keyed VGG16 (torchvision is absent: architecture from oracle/tv_models.py):
    per-image distances, the five levels, and the gradient of 0.5 * sum w.r.t. the predicted image."""
    refshim.install_loss_networks()
    import my_lpips
    t = time.time()
    case, size, B = "lpips64", 64, 3
    pl = my_lpips.PerceptualLoss(model="net-lin", net="vgg", use_gpu=False)
    net = pl.model.net
    spec = spec_of(net)
    lin = {k: v.clone() for k, v in net.state_dict().items() if k.startswith("lin")}
    sd = weights.synth_state_dict("lpips_vgg", [e for e in spec if e[0].startswith("net.")], cases.SEED)
    net.load_state_dict(sd, strict=False)
    net.eval()
    pred, target = cases.image_batch(case + "/pred", B, size), cases.image_batch(case + "/target", B, size)
    out = {"lin/" + k: np_(v) for k, v in lin.items()}
    with torch.enable_grad():
        x = pred.clone().requires_grad_(True)
        val, res = net.forward(target, x, retPerLayer=True)
        out["dist"] = np_(pl(x, target)).reshape(-1)
        for i, r in enumerate(res):
            out[f"level{i}"] = np_(r).reshape(-1)
        loss = pl(x, target).sum() * 0.5
        loss.backward()
        out["loss"], out["d_pred"] = np.array([loss.item()]), np_(x.grad)
    with open(os.path.join(GOLD, "state_specs.json")) as f:
        specs = json.load(f)
    specs["lpips_vgg"] = spec
    with open(os.path.join(GOLD, "state_specs.json"), "w") as f:
        json.dump(specs, f)
    np.savez_compressed(os.path.join(GOLD, "lpips64.npz"), **out)
    print("lpips64.npz: dist", out["dist"], "|d_pred| max %.3e, %d state entries, %.1fs" % (np.abs(out["d_pred"]).max(), len(spec), time.time() - t))


def gen_idloss():
    """The identity term (restoration_train.py:116-117, 242-245): the REFERENCE's Loss.id_loss.IDLoss -- bilinear resize to 112,
    ResNet-101 (256 outputs, eval), L2-normalise, 1 - <z_src, z_out> averaged -- with a keyed ResNet-101 (architecture from
    oracle/tv_models.py; the checkpoint file is written to a temporary directory and read back by the reference's own
    torch.load call).  128x128 inputs so that the resize interpolates."""
    import tempfile
    refshim.install_loss_networks()
    from oracle import tv_models
    t = time.time()
    case, size, B = "idloss128", 128, 2
    z = tv_models.resnet101(num_classes=256)
    spec = spec_of(z)
    sd = weights.synth_state_dict("arcface_resnet101", spec, cases.SEED)
    # A random 101-layer ReLU network maps every image to (almost) the same direction.  Calibrate the BatchNorm running statistics on
    # a keyed batch, as training would have (cumulative average of one pass in train mode), so that the embedding depends on the
    # image; the calibrated statistics are part of the fixture (bn/<name>), everything else regenerates from names.
    z.load_state_dict(sd)
    z.train()
    for m in z.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.reset_running_stats()
            m.momentum = None
    z(torch.nn.functional.interpolate(cases.image_batch(case + "/calib", 8, size), size=112, mode="bilinear"))
    z.eval()
    bn = {k: v.clone() for k, v in z.state_dict().items() if k.endswith("running_mean") or k.endswith("running_var")}
    sd.update(bn)
    real_cuda = torch.nn.Module.cuda
    torch.nn.Module.cuda = lambda self, *a, **k: self       # IDLoss.__init__ moves the network to the GPU (Loss/id_loss.py:15)
    try:
        from Loss.id_loss import IDLoss
        with tempfile.TemporaryDirectory() as d:
            torch.save(sd, os.path.join(d, "arcface.pth"))
            idl = IDLoss(os.path.join(d, "arcface.pth"))
    finally:
        torch.nn.Module.cuda = real_cuda
    pred, target = cases.image_batch(case + "/pred", B, size), cases.image_batch(case + "/target", B, size)
    out = {"bn/" + k: np_(v) for k, v in bn.items()}
    with torch.enable_grad():
        x = pred.clone().requires_grad_(True)
        loss = idl(x, target)
        (loss * 0.1).backward()
        out.update(loss=np.array([loss.item()]), d_pred=np_(x.grad), z_pred=np_(idl.get_id(pred)), z_target=np_(idl.get_id(target)))
    with open(os.path.join(GOLD, "state_specs.json")) as f:
        specs = json.load(f)
    specs["arcface_resnet101"] = spec
    with open(os.path.join(GOLD, "state_specs.json"), "w") as f:
        json.dump(specs, f)
    np.savez_compressed(os.path.join(GOLD, "idloss128.npz"), **out)
    print("idloss128.npz: loss %.5f |d_pred| max %.3e |z| %.3f, %d state entries, %.1fs" % (loss.item(), np.abs(out["d_pred"]).max(),
          np.linalg.norm(out["z_pred"][0]), len(spec), time.time() - t))


def gen_diffuser_train():
    """One iteration of code_diffuser_train.py:153-190 without the two loss networks (pinned on their own: lpips64 / idloss128): the
    REFERENCE's My_DDPM.forward(training=True) (q_sample of the low-quality codes at t = T-1 with a keyed noise, T p_sample steps,
    ldm/ddpm.py:412-421), KDLoss (code_diffuser_train.py:64-90) against keyed target codes, the StyleGAN2 decoder at size 64 on the
    predicted codes (stylegan2_feat_forward_v2: all 10 latents, explicit noise maps) and the scalar
        loss = l_abs + 0.1 * <image, R>
    whose backward reaches the 72 Code_diffuser tensors through the decoder's style path.  Stored: predicted codes, l_kd, l_abs,
    image, loss, a strided sample + norm of every Code_diffuser gradient."""
    t0 = time.time()
    import torch.nn as nn   # (code_diffuser_train.py imports lmdb / tqdm / torchvision at module level: its KDLoss is restated here)
    import torch.nn.functional as F

    class KDLoss(nn.Module):                      # code_diffuser_train.py:64-90, verbatim semantics
        def __init__(self, loss_weight=1.0, temperature=0.15):
            super().__init__()
            self.loss_weight, self.temperature = loss_weight, temperature

        def forward(self, S1_fea, S2_fea):
            dis = ab = 0
            for i in range(len(S1_fea)):
                S2d = F.log_softmax(S2_fea[i] / self.temperature, dim=1)
                S1d = F.softmax(S1_fea[i].detach() / self.temperature, dim=1)
                dis = dis + F.kl_div(S2d, S1d, reduction="batchmean")
                ab = ab + nn.L1Loss()(S2_fea[i], S1_fea[i].detach())
            return self.loss_weight * dis, self.loss_weight * ab

    case, B, T, size = "diffuser_train", 2, 4, 64
    net = Code_diffuser(timesteps=T)
    load_synth(net, "diffuser", cases.SEED)
    ddpm = My_DDPM(denoise=net, linear_start=0.1, linear_end=0.99, timesteps=T)
    g = SG.Generator(size, 512, 8, channel_multiplier=2)
    load_synth(g, "e4e_decoder", cases.SEED)
    n_lat = g.n_latent
    low = cases.tensor(case, "low_latent", (B, 18, 512))
    target = cases.tensor(case, "target", (B, 18, 512))
    q_noise = cases.tensor(case, "q_noise", (B, 18, 512))
    gnoise = cases.noise_list(case, "n", omodels.generator_noise_shapes(size, B))
    R = cases.tensor(case, "R", (B, 3, size, size))
    real_randn_like = torch.randn_like
    torch.randn_like = lambda x, *a, **k: q_noise.to(x.dtype)            # the draw of ldm/ddpm.py:414
    out = {}
    try:
        with torch.enable_grad():
            for p_ in net.parameters():
                p_.requires_grad_(True)
            net.zero_grad()
            pred, lst = ddpm(x=low, condi_in=low, training=True)
            l_kd, l_abs = KDLoss()([target], [lst[-1]])
            img, _ = g([pred[:, :n_lat]], input_is_latent=True, noise=gnoise, return_features=False)
            loss = l_abs + 0.1 * (img * R).sum()
            loss.backward()
    finally:
        torch.randn_like = real_randn_like
    out.update(pred=np_(pred), x_noisy=np_(lst[0]), l_kd=np.array([l_kd.item()]), l_abs=np.array([l_abs.item()]), image=np_(img),
               loss=np.array([loss.item()]))
    names = []
    for n, p_ in net.named_parameters():
        names.append(n)
        out["g/" + n], out["n/" + n] = np_(grad_sample(p_.grad)), np.array([p_.grad.norm().item()])
    out["param_names"] = np.array(names)
    np.savez_compressed(os.path.join(GOLD, "diffuser_train.npz"), **out)
    print("diffuser_train.npz: l_kd %.4f l_abs %.4f loss %.4f, %d parameters, %.1fs" % (l_kd.item(), l_abs.item(), loss.item(), len(names), time.time() - t0))


# ---- f3: the published checkpoints (README.md:49-54) --------------------------------------------------------------------------------
REAL_FILES = {"restorenet": "restoration_net.pt", "diffuser": "code_diffuser.pt", "psp": "style_encoder_decoder.pt"}


def _sha256(path):
    import hashlib
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
    return h.hexdigest()


def real_lq_batch(lq_dir, n):
    """The LQ inputs of the real-checkpoint fixture as the reference's test loader hands them over (dataset.py:436-495:
    ImageFolder_restore_test_no_gt -> RGB, centre crop to a square, LANCZOS resize to 512, [-1, 1]); without a folder: n keyed
    synthetic images (the mechanics can be exercised without faces; parity on faces needs real ones)."""
    if not lq_dir:
        return cases.image_batch("real512", n, 512), ["keyed:real512/%d" % i for i in range(n)]
    from PIL import Image
    names = sorted(f for f in os.listdir(lq_dir) if f.lower().endswith((".png", ".jpg", ".jpeg")))[:n]
    assert len(names) == n, f"need {n} images under {lq_dir}"
    ims = []
    for f in names:
        im = Image.open(os.path.join(lq_dir, f)).convert("RGB")
        w, h = im.size
        m = min(w, h)
        im = im.crop(((w - m) // 2, (h - m) // 2, (w - m) // 2 + m, (h - m) // 2 + m)).resize((512, 512), Image.LANCZOS)
        ims.append(torch.from_numpy(np.asarray(im).copy()).permute(2, 0, 1).float() / 127.5 - 1.0)
    return torch.stack(ims), names


def gen_real512(ckpt_dir=None, lq_dir=None, n=2, out_name="real512.npz", stride=1):
    """SURVEY 8f row 3 (the real-checkpoint path): the REFERENCE's modules with the published weights (restoration_test.py:31-40,
    239-254: restoration_net.pt["g_ema"], code_diffuser.pt["att_mapper"], the pSp dict of style_encoder_decoder.pt) on CPU, n faces,
    every draw pinned by name (oracle.cases) -> the 8-bit images `save_image` would write, stored with the checkpoint hashes.
    tests/test_hip_models.py::test_real_checkpoint_parity loads the same files on the GPU box and asserts <= 1 LSB."""
    from argparse import Namespace
    ckpt_dir = ckpt_dir or os.environ.get("VSPBFR_CKPT_DIR")
    if not ckpt_dir:
        raise SystemExit("real512: pass --ckpt-dir (or VSPBFR_CKPT_DIR) with " + ", ".join(REAL_FILES.values()))
    t0 = time.time()
    case, T = "real512", 4
    paths = {k: os.path.join(ckpt_dir, v) for k, v in REAL_FILES.items()}
    g_ck = torch.load(paths["restorenet"], map_location="cpu")
    d_ck = torch.load(paths["diffuser"], map_location="cpu")
    p_ck = torch.load(paths["psp"], map_location="cpu")
    g = RN.Restoration_net(512, 512, 8, channel_multiplier=2)
    g.load_state_dict(g_ck["g_ema"])
    g.eval()
    net = Code_diffuser(timesteps=T)
    net.load_state_dict(d_ck["att_mapper"])
    net.eval()
    ddpm = My_DDPM(denoise=net, linear_start=0.1, linear_end=0.99, timesteps=T)        # restoration_test.py:31-40
    enc = Encoder4Editing(50, "ir_se", Namespace(input_channel=3, stylegan_size=1024))
    enc.load_state_dict({k[len("encoder."):]: v for k, v in p_ck["state_dict"].items() if k.startswith("encoder.")})
    enc.eval()
    dec = SG.Generator(1024, 512, 8, channel_multiplier=2)
    dec.load_state_dict({k[len("decoder."):]: v for k, v in p_ck["state_dict"].items() if k.startswith("decoder.")})
    dec.eval()
    latent_avg = p_ck["latent_avg"].float()
    lq, names = real_lq_batch(lq_dir, n)
    B = lq.shape[0]
    x256 = torch.nn.functional.interpolate(lq, (256, 256), mode="bilinear")
    codes = (enc(x256) + latent_avg.repeat(B, 1, 1))[:, :18]
    x = cases.tensor(case, "x_T", (B, 18, 512))
    for i in reversed(range(T)):
        x, _ = ddpm.p_sample(x, torch.full((B,), i, dtype=torch.long), codes, clip_denoised=ddpm.clip_denoised)
    pre = x
    gnoise = cases.noise_list(case, "g", omodels.generator_noise_shapes(1024, B))
    img1024, feats = dec([pre], input_is_latent=True, noise=gnoise, return_features=True)
    sample = torch.nn.AdaptiveAvgPool2d((512, 512))(img1024)
    enc_s, dec_s = omodels.restoration_noise_shapes(512, B)
    NOISE_QUEUE.clear()
    NOISE_QUEUE.extend(cases.noise_list(case, "enc", enc_s) + cases.noise_list(case, "dec", dec_s))
    restored = g(lq, feats[:16], pre, [cases.tensor(case, "z", (B, 512))])
    assert not NOISE_QUEUE
    q = lambda t: omodels.save_image_quantize(t)[:, :, ::stride, ::stride].permute(0, 2, 3, 1).contiguous().numpy().astype(np.uint8)  # noqa: E731
    out = {"restored_u8": q(restored), "sample_u8": q(sample), "stride": np.array([stride]), "codes": np_(codes), "pre_latent": np_(pre),
           "names": np.array(names), "sha256": np.array([_sha256(paths[k]) for k in ("restorenet", "diffuser", "psp")]),
           "restored_stats": np.array([restored.mean(), restored.std(), restored.abs().max()], dtype=np.float32)}
    if lq_dir:   # (keyed synthetic inputs regenerate from their names)
        out["lq"] = np_(lq)
    dst = os.path.join(GOLD, out_name)
    np.savez_compressed(dst, **out)
    print(os.path.basename(dst) + ":", B, "images, restored stats", out["restored_stats"], "%.1fs" % (time.time() - t0))


ALL = {"specs": gen_specs, "ops": gen_ops, "layers": gen_layers, "diffuser": gen_diffuser, "ddim": gen_ddim, "restorenet64": gen_restorenet64,
       "generator64": gen_generator64, "encoder": gen_encoder, "pipeline512": gen_pipeline512, "loader": gen_loader, "restorenet64_grad": gen_restorenet64_grad, "discriminator64": gen_discriminator64, "ada": gen_ada, "ada_draws": gen_ada_draws, "lpips": gen_lpips, "idloss": gen_idloss, "diffuser_train": gen_diffuser_train}
OPTIONAL = {"real512": gen_real512}   # needs the published checkpoints: not part of the default set

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=",".join(ALL))
    ap.add_argument("--ckpt-dir", default=None, help="real512: folder with restoration_net.pt, code_diffuser.pt, style_encoder_decoder.pt")
    ap.add_argument("--lq-dir", default=None, help="real512: folder with the LQ faces (default: keyed synthetic images)")
    ap.add_argument("--out-name", default="real512.npz", help="real512: fixture file name under tests/golden")
    ap.add_argument("--n-images", type=int, default=2, help="real512: number of faces")
    ap.add_argument("--stride", type=int, default=1, help="real512: keep every stride-th pixel of the 8-bit images (harness dry runs)")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.set_num_threads(8)
    for k in args.only.split(","):
        if k in OPTIONAL:
            OPTIONAL[k](args.ckpt_dir, args.lq_dir, n=args.n_images, out_name=args.out_name, stride=args.stride)
        else:
            ALL[k]()
