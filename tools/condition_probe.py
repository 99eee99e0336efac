"""Conditioning probe of the synthetic Code_diffuser fixture.  BUILD CONTAINER ONLY (imports the reference).

    python tools/condition_probe.py

For every diffuser case it runs the REAL reference chain (ldm/ddpm.py:400-429 over models/CodeDiffuser.py:86-140) with
the synthetic weights of oracle/weights.py in fp32 and in fp64, and reports
  * |fp32 - fp64| of the end latent  (how far two correct fp32 implementations may differ),
  * the gain of a 1e-5 perturbation of the condition and of x_T (fp64 finite difference),
  * the magnitude of the end latent.
The fixture is usable for a 1e-3 end-to-end bound only if the first figure is <= 1e-5 and the gains are O(1).
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import refshim  # noqa: E402

refshim.install()

from oracle import cases, weights  # noqa: E402
from models.CodeDiffuser import Code_diffuser  # noqa: E402
from ldm.ddpm import My_DDPM  # noqa: E402

torch.set_grad_enabled(False)
SCALES = []  # [(regex, factor)] experimental rescales on top of oracle/weights.py (command line: regex=factor ...)


def chain(ddpm, cond, x, T):
    B = x.shape[0]
    for i in reversed(range(T)):
        x, _ = ddpm.p_sample(x, torch.full((B,), i, dtype=torch.long), cond, clip_denoised=ddpm.clip_denoised)
    return x


def build(T, ls, le, dtype):
    net = Code_diffuser(timesteps=T)
    spec = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in net.state_dict().items()]
    sd = weights.synth_state_dict("diffuser", spec, cases.SEED)
    for pat, sc in SCALES:
        for k in sd:
            if re.search(pat, k):
                sd[k] = sd[k] * sc
    net.load_state_dict(sd)
    ddpm = My_DDPM(denoise=net, linear_start=ls, linear_end=le, timesteps=T).eval()
    return ddpm.to(dtype)


def probe(name, B, T, ls, le, cond=None):
    cond = cases.tensor(name, "cond", (B, 18, 512)) if cond is None else cond
    x_T = cases.tensor(name, "x_T", (B, 18, 512))
    d32, d64 = build(T, ls, le, torch.float32), build(T, ls, le, torch.float64)
    y32 = chain(d32, cond, x_T, T)
    c64, x64 = cond.double(), x_T.double()
    y64 = chain(d64, c64, x64, T)
    g = torch.Generator().manual_seed(1)
    dc = torch.randn(cond.shape, generator=g, dtype=torch.float64)
    dc = dc / dc.abs().max() * 1e-5
    yc = chain(d64, c64 + dc, x64, T)
    yx = chain(d64, c64, x64 + dc, T)
    print(f"{name:14s} T={T:3d} |y|max {y64.abs().max():6.2f} std {y64.std():5.2f}   fp32-fp64 {(y32.double() - y64).abs().max():.2e}"
          f"   gain(cond) {(yc - y64).abs().max() / 1e-5:8.2f}   gain(x_T) {(yx - y64).abs().max() / 1e-5:8.3f}")


if __name__ == "__main__":
    torch.set_num_threads(8)
    for a in sys.argv[1:]:
        pat, sc = a.rsplit("=", 1)
        SCALES.append((pat, float(sc)))
    probe("ddpm_T4", 2, 4, 0.1, 0.99)
    probe("ddpm_T10", 2, 10, 1e-4, 2e-2)
    probe("ddim_T50_S25", 2, 50, 1e-4, 2e-2)
    # the pipeline case conditions the chain on the e4e encoder's codes (tests/golden/pipeline512.npz, stage A of the same run)
    import numpy as np
    codes = torch.from_numpy(np.load(os.path.join(ROOT, "tests", "golden", "pipeline512.npz"))["codes"])
    print("codes: std %.3f absmax %.3f  token-mean std %.3f  across-token std %.3f" % (
        codes.std(), codes.abs().max(), codes.mean(1).std(), (codes - codes.mean(1, keepdim=True)).std()))
    probe("pipeline512", 1, 4, 0.1, 0.99, cond=codes)
    probe("pipeline512", 1, 50, 1e-4, 2e-2, cond=codes)


def floor(name="ddpm_T4", B=2, T=4):
    """fp32 vs fp64 of ONE denoiser call and of each TACC block on identical inputs: the floor no chain can beat."""
    d32, d64 = build(T, 0.1, 0.99, torch.float32), build(T, 0.1, 0.99, torch.float64)
    cond = cases.tensor(name, "cond", (B, 18, 512))
    x = chain(d64, cond.double(), cases.tensor(name, "x_T", (B, 18, 512)).double(), T - 1)  # the input of the last step
    t = torch.zeros(B, dtype=torch.long)
    y64, y32 = d64.model(x, cond.double(), t), d32.model(x.float(), cond, t)
    print(f"one call at t=0: |y|max {y64.abs().max():.2f}  fp32-fp64 {(y32.double() - y64).abs().max():.2e}")
    step = torch.zeros(B, 18, 1, dtype=torch.float64)
    h = x
    for i, (m64, m32) in enumerate(zip(d64.model.att_mapper, d32.model.att_mapper)):
        o64, o32 = m64(h, cond.double(), step), m32(h.float(), cond, step.float())
        print(f"  block {i}: fp32-fp64 on the same input {(o32.double() - o64).abs().max():.2e}  |out|max {o64.abs().max():.2f}")
        h = o64
