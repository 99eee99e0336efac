"""What the "hidden" side stream costs (review r5 item 7).  One process, one box, same weights and inputs:

  full       : RestorationPipeline.run_batches -- stages A + B of batch i+1 on the side stream under C + D of batch i (what bench.py times)
  main_only  : stages C + D alone, latents pre-computed once (the main-stream floor)
  ab_only    : stages A + B alone, serial (what the side stream has to hide)
  serial     : A + B + C + D on one stream

`full - main_only` is what the 8-9 ms of stage-A convolutions and the 600 chain launches really cost underneath; with --preset c3 the same
for BASELINE configs[2], plus stage A on the fp32 kernels (pipe.encoder_fp32): throughput and the latent / image deviation from the fp32 run.

usage: bench_hidden_cost.py [--preset c2|c3] [--steps K]  -> one JSON line
"""
import argparse
import copy
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--preset", default="c2")
ap.add_argument("--steps", type=int, default=12)
args = ap.parse_args()
cfg = bench.PRESETS[args.preset]
B, K = cfg["batch"], args.steps
dev = torch.device("cuda", 0)
from vspbfr_amd import hip_ops  # noqa: E402

pipe = bench.build_pipeline(dev, cfg["timesteps"], True, 123)
if cfg.get("sampler") == "ddim":
    from vspbfr_amd.ddim import DDIMSampler
    sampler, S = DDIMSampler(pipe.diffusion, device=dev), cfg["ddim_steps"]

    class _DDIM(torch.nn.Module):
        def forward(self, x=None, condi_in=None, training=False, x_T=None):
            return sampler.sample(S=S, batch_size=condi_in.shape[0], shape=18 * 512, conditioning=condi_in, eta=0.0, verbose=False, x_T=x_T)[0]
    pipe.diffusion = _DDIM()
bf = cfg.get("conv_dtype") == "bf16"
hip_ops.BF16_CONV = bf
pipe.act_bf16 = bool(cfg.get("act_bf16"))
lq = hip_ops.keyed_fill([(B, 3, 512, 512)], [hip_ops.SEG_LQ], 123, 0, dist="uniform", device=dev)[0]


def timed(fn, n=K):
    fn(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def full(n):
    for o in pipe.run_batches([(lq, i * B) for i in range(n)]):
        pass


def serial(n):
    for i in range(n):
        pipe(lq, image_index0=i * B)


with torch.no_grad():
    lat, pre = pipe.encode(lq, image_index0=0)
    lat, pre = lat.clone(), pre.clone()

    def main_only(n):
        for i in range(n):
            pipe.decode(lq, lat, pre, image_index0=i * B)

    def ab_only(n):
        for i in range(n):
            pipe.encode(lq, image_index0=i * B)

    out = {"preset": args.preset, "batch": B, "steps": K}
    for name, fn in (("full", full), ("main_only", main_only), ("ab_only", ab_only), ("serial", serial), ("full_again", full)):
        out[name + "_ms"] = round(timed(fn), 2)
    out["hidden_cost_ms"] = round(min(out["full_ms"], out["full_again_ms"]) - out["main_only_ms"], 2)
    out["img_per_s_full"] = round(B / min(out["full_ms"], out["full_again_ms"]) * 1e3, 1)
    out["img_per_s_main_only"] = round(B / out["main_only_ms"] * 1e3, 1)
    if bf:
        # stage A on the fp32 kernels: throughput, and how far latents / images move towards the all-fp32 run
        ref_bf = pipe(lq, image_index0=0)
        r_bf = {k: ref_bf[k].float().clone() for k in ("latent", "pre_latent", "restored")}
        pipe.encoder_fp32 = True
        out["full_encoder_fp32_ms"] = round(timed(full), 2)
        out["img_per_s_encoder_fp32"] = round(B / out["full_encoder_fp32_ms"] * 1e3, 1)
        mix = pipe(lq, image_index0=0)
        r_mix = {k: mix[k].float().clone() for k in ("latent", "pre_latent", "restored")}
        pipe.encoder_fp32 = False
        hip_ops.BF16_CONV = False
        pipe.act_bf16 = False
        f32 = pipe(lq, image_index0=0)
        # (bench.build_pipeline's networks are random-init WITHOUT the fixture's rescale: their images are not unit-scale -- deviations are given
        #  relative to the fp32 image's standard deviation; the unit-scale figures are test_config_c3_full_size's, profiles/parity_c3_*_r06.json)
        sigma = float(f32["restored"].float().std())
        out["restored_sigma_fp32"] = round(sigma, 4)
        for tag, r in (("bf16_all", r_bf), ("bf16_encoder_fp32", r_mix)):
            d = (r["restored"] - f32["restored"].float())
            out[tag] = {"codes_max": round(float((r["latent"] - f32["latent"].float()).abs().max()), 6),
                        "pre_latent_max": round(float((r["pre_latent"] - f32["pre_latent"].float()).abs().max()), 6),
                        "restored_rms_over_sigma": round(float(d.pow(2).mean().sqrt()) / sigma, 5),
                        "restored_max_over_sigma": round(float(d.abs().max()) / sigma, 5)}
print(json.dumps(out))
