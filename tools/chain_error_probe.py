"""Spread of the free-running T = 50 chain error (HIP sampler vs the CPU oracle on the same codes and x_T) over different inputs:
what bound tests/test_hip_models.py::test_config_c2_full_size can assert.  usage: python tools/chain_error_probe.py [n_batches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_hip_models as M
from oracle import cases, weights, device_rng as R
from vspbfr_amd import hip_ops as H
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
B, T, seed = 8, 50, 2025
pipe = M.build_pipeline(T=T, linear_start=1e-4, linear_end=2e-2, with_sample=False)
pipe.noise_seed = seed
sd = weights.synth_state_dict("diffuser", weights.load_specs()["diffuser"], cases.SEED)
for j in range(n):
    i0 = 16 + 8 * j
    lq = H.keyed_fill([(B, 3, 512, 512)], [H.SEG_LQ], seed, i0, dist="uniform")[0]
    out = pipe(lq, image_index0=i0)
    x_T = torch.from_numpy(R.keyed_fill((B, 18, 512), H.SEG_XT, seed, i0))
    ref = M._oracle_chain(sd, out["latent"].cpu(), x_T, T)
    d = (out["pre_latent"].cpu() - ref).abs()
    per_img = d.reshape(B, -1).max(1).values
    print(f"images {i0}..{i0 + B - 1}: max {float(d.max()):.2e}  per image " + " ".join(f"{float(v):.1e}" for v in per_img) + f"  |latent|max {float(ref.abs().max()):.1f}")
