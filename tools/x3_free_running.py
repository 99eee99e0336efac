import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from oracle import cases, models as OM
import test_hip_models as T
from vspbfr_amd import hip_ops
case, B = "pipeline512", 1
dev = T.dev
g = dict(np.load("/root/repo/tests/golden/pipeline512.npz"))
pipe = T.build_pipeline()
lq = cases.image_batch(case, B, 512)
gno = [dev(n) for n in cases.noise_list(case, "g", OM.generator_noise_shapes(1024, B))]
enc_s, dec_s = OM.restoration_noise_shapes(512, B)
z = [dev(cases.tensor(case, "z", (B, 512)))]
en = [dev(n) for n in cases.noise_list(case, "enc", enc_s)]
dn = [dev(n) for n in cases.noise_list(case, "dec", dec_s)]
for mode, encfp32 in ((False, True), ("x3", True), ("x3", False), (True, True)):
    hip_ops.BF16_CONV = mode; pipe.encoder_fp32_under_x3 = encfp32
    out = pipe(dev(lq), z=z, x_T=dev(cases.tensor(case, "x_T", (B, 18, 512))), gen_noise=gno, enc_noise=en, dec_noise=dn)
    r = out["restored"]
    print(mode, encfp32, "codes", T.maxerr(out["latent"], g["codes"]), "pre_latent", T.maxerr(out["pre_latent"], g["pre_latent"]),
          "restored_sub", T.maxerr(r[:, :, ::8, ::8], g["restored_sub"]), "sample_sub", T.maxerr(out["style_sample"][:, :, ::8, ::8], g["sample_sub"]))
