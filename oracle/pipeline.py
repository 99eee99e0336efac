"""CPU statement of the whole restoration step A -> B -> C -> D (reference restoration_test.py:125-131) on synthetic
weights (TEST INFRASTRUCTURE; also bench.py's `cpu_baseline` leg)."""
import torch

from . import cases, models, weights


def synth_checkpoints(seed=cases.SEED, size=512):
    """The four state dicts of the path, from the name-keyed synthesiser."""
    specs = weights.load_specs()
    return {
        "encoder": weights.synth_state_dict("e4e_encoder", specs["e4e_encoder"], seed),
        "decoder": weights.synth_state_dict("e4e_decoder", specs["e4e_decoder1024"], seed),
        "latent_avg": weights.synth_tensor("e4e_decoder", "latent_avg", (18, 512), "float32", seed),
        "diffuser": weights.synth_state_dict("diffuser", specs["diffuser"], seed),
        "restorenet": weights.synth_state_dict("restorenet", specs[f"restorenet{size}"], seed),
    }


def psp_checkpoint(ck):
    """The pSp checkpoint dict layout E4e_embedding loads (reference Loss/e4e_embedding.py:85-88)."""
    sd = {"encoder." + k: v for k, v in ck["encoder"].items()}
    sd.update({"decoder." + k: v for k, v in ck["decoder"].items()})
    return {"state_dict": sd, "latent_avg": ck["latent_avg"],
            "opts": {"encoder_type": "Encoder4Editing", "stylegan_size": 1024, "start_from_latent_avg": True}}


def draw_inputs(case, B, size=512, gen_size=1024):
    """Every tensor the step consumes, keyed by name: LQ batch, z, x_T and the per-layer noise maps."""
    enc_s, dec_s = models.restoration_noise_shapes(size, B)
    return {
        "lq": cases.image_batch(case, B, size),
        "z": cases.tensor(case, "z", (B, 512)),
        "x_T": cases.tensor(case, "x_T", (B, 18, 512)),
        "gen_noise": cases.noise_list(case, "g", models.generator_noise_shapes(gen_size, B)),
        "enc_noise": cases.noise_list(case, "enc", enc_s),
        "dec_noise": cases.noise_list(case, "dec", dec_s),
    }


@torch.no_grad()
def restore(ck, inp, timesteps=4, linear_start=0.1, linear_end=0.99, size=512, timings=None):
    import time
    t0 = time.perf_counter()
    codes = models.get_w_plus(ck["encoder"], inp["lq"], ck["latent_avg"], p="")
    t1 = time.perf_counter()
    pre = models.ddpm_sample(ck["diffuser"], codes, inp["x_T"], timesteps, linear_start, linear_end)
    t2 = time.perf_counter()
    sample, feats = models.stylegan_feats(ck["decoder"], 1024, size, pre, inp["gen_noise"])
    t3 = time.perf_counter()
    restored = models.restoration_net(ck["restorenet"], size, inp["lq"], feats, pre, [inp["z"]], inp["enc_noise"],
                                      inp["dec_noise"])
    t4 = time.perf_counter()
    if timings is not None:
        timings.update(encoder=t1 - t0, diffuser=t2 - t1, prior_decoder=t3 - t2, restorenet=t4 - t3)
    return {"restored": restored, "style_sample": sample, "latent": codes, "pre_latent": pre}
