"""Restored-image delta of the bf16-kernel configuration against the fp32 path on the same inputs and the same noise
(torch.manual_seed before each run: the device RNG draw order does not depend on the conv kernels)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vspbfr_amd import hip_ops
dev = torch.device("cuda", 0)
B, T = int(os.environ.get("B", 2)), int(os.environ.get("T", 4))
pipe = bench.build_pipeline(dev, T, True)
lq = torch.rand(B, 3, 512, 512, device=dev) * 2 - 1
def run(bf, lat=None):
    hip_ops.BF16_CONV = bf
    torch.manual_seed(7)
    with torch.no_grad():
        if lat is None:
            o = pipe(lq)
        else:
            o = pipe.decode(lq, lat[0], lat[1])
    torch.cuda.synchronize()
    return o
ref = run(False)
full = run(True)
lat = (ref["latent"], ref["pre_latent"])
ref_cd = run(False, lat)
fed = run(True, lat)   # stages C + D in bf16 on the fp32 path's latents, same noise draws as ref_cd
d = lambda a, b: (a - b).abs().max().item()
rms = lambda a, b: (a - b).pow(2).mean().sqrt().item()
out = {"B": B, "T": T,
       "restored_absmax": ref["restored"].abs().max().item(), "restored_std": ref["restored"].std().item(),
       "free_running": {"latent": d(full["latent"], ref["latent"]), "pre_latent": d(full["pre_latent"], ref["pre_latent"]),
                        "restored_max": d(full["restored"], ref["restored"]), "restored_rms": rms(full["restored"], ref["restored"]),
                        "style_sample_max": d(full["style_sample"], ref["style_sample"])},
       "C+D on the fp32 latents": {"restored_max": d(fed["restored"], ref_cd["restored"]), "restored_rms": rms(fed["restored"], ref_cd["restored"]),
                                   "style_sample_max": d(fed["style_sample"], ref_cd["style_sample"]),
                                   "style_sample_rms": rms(fed["style_sample"], ref_cd["style_sample"])}}
print(json.dumps(out))
