"""Driver smoke test: one small invocation of the hot path on cuda:0, checked against the CPU oracle.

Uses the size-64 Restoration_net / size-64 StyleGAN2 prior (512 channels at every level, 102 M + 28 M parameters) and a
T=4 Code_diffuser chain so that the oracle finishes in seconds; the full 512^2 path is covered by tests/ (-m gpu).
This is the ONLY module under vspbfr_amd/ that touches oracle/ -- as the checker, per the tier rules."""
import time

import torch


def run_smoke():
    from oracle import cases, models as OM, weights
    from . import _lib
    from .diffusion import Code_diffuser, My_DDPM
    from .e4e import Generator
    from .restorenet import Restoration_net

    assert torch.cuda.is_available(), "smoke() needs a GPU"
    assert _lib.lib.vsp_device_count() >= 1, _lib.last_error()
    dev = "cuda:0"
    torch.cuda.set_device(0)
    t0 = time.time()
    specs = weights.load_specs()
    B, size, T = 1, 64, 4
    with torch.no_grad():
        # B: latent denoising chain
        sd_d = weights.synth_state_dict("diffuser", specs["diffuser"], cases.SEED)
        net = Code_diffuser(timesteps=T)
        net.load_state_dict(sd_d)
        ddpm = My_DDPM(denoise=net.to(dev).eval(), linear_start=0.1, linear_end=0.99, timesteps=T).to(dev)
        cond, x_T = cases.tensor("smoke", "cond", (B, 18, 512)), cases.tensor("smoke", "x_T", (B, 18, 512))
        pre = ddpm(x=cond.to(dev), condi_in=cond.to(dev), training=False, x_T=x_T.to(dev))
        pre_ref = OM.ddpm_sample(sd_d, cond, x_T, T, 0.1, 0.99)
        e_b = (pre.cpu() - pre_ref).abs().max().item()
        # C: StyleGAN2 prior with feature taps (fed with the oracle's latent so the stages are checked independently)
        sd_g = weights.synth_state_dict("e4e_decoder", specs["e4e_decoder64"], cases.SEED)
        gen = Generator(size, 512, 8)
        gen.load_state_dict(sd_g)
        gen = gen.to(dev).eval()
        gnoise = cases.noise_list("smoke", "g", OM.generator_noise_shapes(size, B))
        lat = pre_ref[:, :10].contiguous()
        img, feats = gen([lat.to(dev)], input_is_latent=True, noise=[n.to(dev) for n in gnoise], return_features=True)
        img_ref, feats_ref = OM.stylegan_generator(sd_g, size, lat, gnoise)
        e_c = max((img.cpu() - img_ref).abs().max().item(), max((a.cpu() - b).abs().max().item() for a, b in zip(feats, feats_ref)))
        # D: Restoration_net
        sd_r = weights.synth_state_dict("restorenet", specs["restorenet64"], cases.SEED)
        rn = Restoration_net(size, 512, 8)
        rn.load_state_dict(sd_r)
        rn = rn.to(dev).eval()
        lq, z = cases.image_batch("smoke", B, size), cases.tensor("smoke", "z", (B, 512))
        enc_s, dec_s = OM.restoration_noise_shapes(size, B)
        en, dn = cases.noise_list("smoke", "enc", enc_s), cases.noise_list("smoke", "dec", dec_s)
        out = rn(lq.to(dev), [f.to(dev) for f in feats_ref], pre_ref.to(dev), [z.to(dev)], enc_noise=[n.to(dev) for n in en],
                 dec_noise=[n.to(dev) for n in dn])
        ref = OM.restoration_net(sd_r, size, lq, feats_ref, pre_ref, [z], en, dn)
        e_d = (out.cpu() - ref).abs().max().item()
    torch.cuda.synchronize()
    print(f"smoke: max|d| vs oracle  chain={e_b:.2e}  prior={e_c:.2e}  restorenet={e_d:.2e}  ({time.time() - t0:.1f}s, "
          f"lib={_lib.LIB_PATH})")
    assert e_b < 2e-4 and e_c < 2e-4 and e_d < 2e-4, (e_b, e_c, e_d)
    assert torch.isfinite(out).all()
