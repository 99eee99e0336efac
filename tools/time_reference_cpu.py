"""Time the REAL reference (/root/reference) on this container's CPU cores.  BUILD CONTAINER ONLY.

    python tools/time_reference_cpu.py [--threads 8] [--cases c1,b1t50]

The per-batch body of restoration_test.py:125-131 (A get_w_plus -> B diffusion -> C get_stylegan_feats -> D generator) driven
through the reference's own modules with random-init weights of the real shapes (restoration_test.py itself cannot be imported
here: it needs torchvision; the four calls are restated as tools/make_golden.py::gen_pipeline512 does).  1 warm-up + N timed
batches per case, per-stage split, img/s.  Cases: c1 = BASELINE.json configs[0] (B = 4, T = 10 DDPM); b1t50 = one image at
T = 50 (the sample bench.py's cpu_baseline leg times through the oracle on the GPU box's host); c2 = configs[1] itself (B = 8, T = 50)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import refshim  # noqa: E402

refshim.install()

import models.RestoreNet as RN  # noqa: E402  (reference)
from models.CodeDiffuser import Code_diffuser  # noqa: E402
from ldm.ddpm import My_DDPM  # noqa: E402
import e4e.models.stylegan2.model as SG  # noqa: E402
from e4e.models.encoders.psp_encoders import Encoder4Editing  # noqa: E402

torch.set_grad_enabled(False)


def run_case(B, T, ls, le, batches):
    from argparse import Namespace
    torch.manual_seed(0)
    enc = Encoder4Editing(50, "ir_se", Namespace(input_channel=3, stylegan_size=1024)).eval()
    dec = SG.Generator(1024, 512, 8, channel_multiplier=2).eval()
    net = Code_diffuser(timesteps=T).eval()
    ddpm = My_DDPM(denoise=net, linear_start=ls, linear_end=le, timesteps=T).eval()
    gen = RN.Restoration_net(512, 512, 8).eval()
    latent_avg = 0.1 * torch.randn(18, 512)
    pool = torch.nn.AdaptiveAvgPool2d((512, 512))
    lq = torch.rand(B, 3, 512, 512) * 2 - 1
    st = {"encoder": 0.0, "diffuser": 0.0, "prior_decoder": 0.0, "restorenet": 0.0}
    for it in range(batches + 1):
        t0 = time.perf_counter()
        x256 = torch.nn.functional.interpolate(lq, (256, 256), mode="bilinear")          # Loss/e4e_embedding.py:91-100
        codes = (enc(x256) + latent_avg.repeat(B, 1, 1))[:, :18]                          # e4e/models/psp.py:145-165
        t1 = time.perf_counter()
        pre = ddpm(x=codes, condi_in=codes, training=False)                               # ldm/ddpm.py:420-429
        t2 = time.perf_counter()
        img, feats = dec([pre], input_is_latent=True, randomize_noise=True, return_features=True)   # e4e/models/psp.py:235-248
        sample, feats = pool(img), feats[:16]
        t3 = time.perf_counter()
        restored = gen(lq, feats, pre, [torch.randn(B, 512)])                             # restoration_test.py:131
        t4 = time.perf_counter()
        assert torch.isfinite(restored).all() and restored.shape == (B, 3, 512, 512)
        if it:  # the first batch warms the allocator and oneDNN's primitive cache
            for k, v in zip(st, (t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
                st[k] += v / batches
    total = sum(st.values())
    return {"batch": B, "timesteps": T, "batches_timed": batches, "s_per_batch": round(total, 2), "img_per_s": round(B / total, 4),
            "stage_seconds_per_batch": {k: round(v, 2) for k, v in st.items()}}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ap.add_argument("--cases", default="c1,b1t50")
    ap.add_argument("--batches", type=int, default=3)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_reference_cpu_timing.json"))
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    # c2 = BASELINE.json configs[1], the configuration the metric is quoted on (B = 8, T = 50)
    cases = {"c1": (4, 10, 1e-4, 2e-2), "b1t50": (1, 50, 1e-4, 2e-2), "c2": (8, 50, 1e-4, 2e-2)}
    rep = {"what": "the reference's own modules (torch %s CPU) in the build container" % torch.__version__, "threads": a.threads,
           "host_cpus": os.cpu_count()}
    for c in a.cases.split(","):
        rep[c] = run_case(*cases[c], a.batches)
        print(c, rep[c], flush=True)
    with open(a.out, "w") as f:
        json.dump(rep, f, indent=1)
