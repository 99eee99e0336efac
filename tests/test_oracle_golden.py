"""Pins the CPU oracle (oracle/) to the real reference: every golden vector under tests/golden/ was produced by
tools/make_golden.py importing /root/reference in the build container; here the oracle recomputes them from the same
name-keyed inputs.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import cases, models, ops, weights

torch.set_grad_enabled(False)


def close(a, b, rtol=2e-5, atol=2e-5):
    a = a.numpy() if isinstance(a, torch.Tensor) else a
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max()
    tol = atol + rtol * np.abs(b).max()
    assert err <= tol, f"max|d|={err:.3e} tol={tol:.3e}"


@pytest.mark.parametrize("name", list(cases.LRELU_CASES))
def test_fused_leaky_relu(golden, name):
    x, b = cases.lrelu_inputs(name)
    close(ops.fused_leaky_relu(x, b), golden("ops")[name], 1e-6, 1e-6)


@pytest.mark.parametrize("name", list(cases.FIR_CASES))
def test_upfirdn2d(golden, name):
    x, k, up, down, pad = cases.fir_inputs(name)
    close(ops.upfirdn2d(x, k, up, down, pad), golden("ops")[name], 1e-6, 1e-6)


def _sd(case, names_shapes):
    return cases.module_weights(case, names_shapes)


def _modconv_shapes(cin, cout, k, sdim, blur):
    s = [("weight", (1, cout, cin, k, k))]
    if blur:
        s.append(("blur.kernel", (4, 4)))
    return s + [("modulation.weight", (cin, sdim)), ("modulation.bias", (cin,))]


@pytest.mark.parametrize("name", list(cases.MODCONV_CASES))
def test_modulated_conv(golden, name):
    kind, cin, cout, k, sdim, xs, extra = cases.MODCONV_CASES[name]
    sd = _sd(name, _modconv_shapes(cin, cout, k, sdim, kind != "same"))
    x, style = cases.tensor(name, "x", xs), cases.tensor(name, "style", (xs[0], sdim))
    mod = ops.equal_linear(style, sd["modulation.weight"], sd["modulation.bias"])
    y = ops.modulated_conv(x, sd["weight"], mod, demodulate=extra.get("demodulate", True), mode=kind,
                           blur_kernel=sd.get("blur.kernel"))
    close(y, golden("layers")[name])


@pytest.mark.parametrize("name", list(cases.DILCONV_CASES))
def test_dilated_modulated_conv(golden, name):
    cin, cout, xs, d = cases.DILCONV_CASES[name]
    sd = _sd(name, [("weight", (1, cout, cin, 3, 3))])
    x, style = cases.tensor(name, "x", xs), cases.tensor(name, "style", (xs[0], cin)) * 0.5 + 1.0
    close(ops.modulated_conv(x, sd["weight"], style, dilation=d), golden("layers")[name])


def test_smart_layer(golden):
    name = "smart_16"
    cin, cout, sdim, xs = cases.SMART_CASES[name]
    shapes = [(f"ModulatedConv2ds.{i}.weight", (1, cout // 4, cin, 3, 3)) for i in range(4)]
    shapes += [("modulation.weight", (cin, sdim)), ("modulation.bias", (cin,)), ("fusion.0.weight", (cout, cout, 3, 3)),
               ("fusion.1.bias", (cout,)), ("noise.weight", (1,)), ("activate.bias", (cout,))]
    sd = _sd(name, shapes)
    x, style = cases.tensor(name, "x", xs), cases.tensor(name, "style", (xs[0], sdim))
    noise = cases.tensor(name, "noise", (xs[0], 1, xs[2], xs[3]))
    close(models.smart_layer(sd, "", x, style, noise), golden("layers")[name])


@pytest.mark.parametrize("name", list(cases.LARGECONV_CASES))
def test_large_conv_layer(golden, name):
    cin, cout, k, xs = cases.LARGECONV_CASES[name]
    shapes = [(f"dilated_convs.{i}.weight", (cout // 4, cin, k, k)) for i in range(4)]
    shapes += [("fusion.0.weight", (cout, cout, 1, 1)), ("fusion.1.bias", (cout,)), ("activate.bias", (cout,))]
    sd = _sd(name, shapes)
    close(models.large_conv_layer(sd, "", cases.tensor(name, "x", xs)), golden("layers")[name])


@pytest.mark.parametrize("name", list(cases.DIFFUSER_CASES))
def test_diffuser_and_ddpm(golden, name):
    B, T, ls, le = cases.DIFFUSER_CASES[name]
    sd = weights.synth_state_dict("diffuser", weights.load_specs()["diffuser"], cases.SEED)
    cond, x_T = cases.diffuser_inputs(name)
    g = golden("diffuser")
    _, _, c1, c2 = models.ddpm_schedule(T, ls, le)
    np.testing.assert_array_equal(c1.numpy(), g[name + "/coef1"])
    np.testing.assert_array_equal(c2.numpy(), g[name + "/coef2"])
    t = torch.full((B,), T - 1, dtype=torch.long)
    close(models.code_diffuser(sd, x_T, cond, t, T), g[name + "/x0_first"])
    close(models.ddpm_sample(sd, cond, x_T, T, ls, le), g[name + "/final"], 5e-5, 5e-5)


def test_ddpm_T4_schedule_matches_survey():
    # beta = [.1, .2943, .5910, .99]; c1 = [1, .7652, .6363, .5059]; c2 = [0, .2302, .3153, .0742]
    b, _, c1, c2 = models.ddpm_schedule(4, 0.1, 0.99)
    np.testing.assert_allclose(b.numpy(), [0.1, 0.2943, 0.5910, 0.99], atol=5e-4)
    np.testing.assert_allclose(c1.numpy(), [1.0, 0.7652, 0.6363, 0.5059], atol=5e-4)
    np.testing.assert_allclose(c2.numpy(), [0.0, 0.2302, 0.3153, 0.0742], atol=5e-4)


def test_restorenet64(golden):
    size, B, case = 64, 1, "restorenet64"
    sd = weights.synth_state_dict("restorenet", weights.load_specs()["restorenet64"], cases.SEED)
    imgs = cases.image_batch(case, B, size)
    enc_s, dec_s = models.restoration_noise_shapes(size, B)
    chans = {4: 512, 8: 512, 16: 512, 32: 512, 64: 512}
    de_feats = [cases.tensor(case, f"de_feat{k}", (B, chans[2 ** (k + 2)], 2 ** (k + 2), 2 ** (k + 2)), 0.5) for k in range(5)]
    pre = cases.tensor(case, "pre_styles", (B, 18, 512))
    z = cases.tensor(case, "z", (B, 512))
    out = models.restoration_net(sd, size, imgs, de_feats, pre, [z], cases.noise_list(case, "enc", enc_s),
                                 cases.noise_list(case, "dec", dec_s))
    close(out, golden(case)["image"], 1e-4, 1e-4)


def test_generator64(golden):
    sd = weights.synth_state_dict("e4e_decoder", weights.load_specs()["e4e_decoder64"], cases.SEED)
    B = 2
    latent = cases.tensor("generator64", "latent", (B, 10, 512))
    noise = cases.noise_list("generator64", "n", models.generator_noise_shapes(64, B))
    img, feats = models.stylegan_generator(sd, 64, latent, noise)
    g = golden("generator64")
    close(img, g["image"], 1e-4, 1e-4)
    for i, f in enumerate(feats):
        close(cases.feat_sample(f), g[f"feat{i}"], 1e-4, 1e-4)


def test_encoder4editing(golden):
    sd = weights.synth_state_dict("e4e_encoder", weights.load_specs()["e4e_encoder"], cases.SEED)
    x = cases.image_batch("encoder", 1, 256)
    close(models.encoder4editing(sd, x), golden("encoder")["codes"], 1e-4, 1e-4)


def test_save_image_quantizer():
    x = torch.tensor([-2.0, -1.0, -0.5, 0.0, 0.5, 1.0, 3.0, 0.999])
    np.testing.assert_array_equal(models.save_image_quantize(x).numpy(), [0, 0, 64, 128, 191, 255, 255, 255])


def test_pipeline512(golden):
    """The whole path A+B+C+D at 512^2 (T=4 as shipped) through oracle/pipeline.py against the reference run."""
    from oracle import pipeline as OP
    ck = OP.synth_checkpoints()
    inp = OP.draw_inputs("pipeline512", 1)
    out = OP.restore(ck, inp, timesteps=4, linear_start=0.1, linear_end=0.99)
    g = golden("pipeline512")
    close(out["latent"], g["codes"], 1e-4, 1e-4)
    close(out["pre_latent"], g["pre_latent"], 1e-4, 1e-4)
    close(out["style_sample"][:, :, ::8, ::8], g["sample_sub"], 2e-4, 2e-4)
    close(out["restored"][:, :, ::8, ::8], g["restored_sub"], 2e-4, 2e-4)
    close(out["restored"][:, :, 200:264, 200:264], g["restored_crop"], 2e-4, 2e-4)


def test_ddim(golden):
    """DDIM S=25 on T=50 (BASELINE config 3) against the reference's DDIMSampler run (tools/make_golden.py::gen_ddim)."""
    name, B, T, S = cases.DDIM_CASE
    sd = weights.synth_state_dict("diffuser", weights.load_specs()["diffuser"], cases.SEED)
    cond, x_T = cases.diffuser_inputs(name)
    g = golden("ddim")
    _, ac, _, _ = models.ddpm_schedule(T)
    steps, a, a_prev, _ = models.ddim_schedule(ac, S)
    np.testing.assert_array_equal(steps, g["ddim_timesteps"])
    np.testing.assert_allclose(a, g["ddim_alphas"], rtol=1e-7)
    np.testing.assert_allclose(a_prev, g["ddim_alphas_prev"], rtol=1e-7)
    close(models.ddim_sample(sd, cond, x_T, T, S), g["final"], 1e-5, 1e-5)
