import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
def b16(t): return t.to(torch.bfloat16)
k = torch.tensor([1., 3., 3., 1.]); k = (k[:, None] * k[None, :]); k = (k / k.sum() * 4).cuda()
junk = []
VAR = os.environ.get("VAR", "all")
for it in range(60):
    torch.manual_seed(it)
    B, C, Hh, Ww, pad = 2, 5, 33, 129, (1, 1)
    if it % 3 == 1: junk.append(torch.full((1000 + 37 * it,), float("nan"), device="cuda"))
    x = b16(torch.randn(B, C, Hh, Ww, device="cuda"))
    oh, ow = Hh - 1, Ww - 1
    nz = torch.randn(B, 1, oh, ow, device="cuda"); nw = torch.full((1,), 0.3, device="cuda"); ab = torch.randn(C, device="cuda")
    r1 = b16(torch.randn(B, C, oh, ow, device="cuda")); r2 = b16(torch.randn(B, C, oh, ow, device="cuda"))
    kw = {"all": dict(noise=nz, noise_w=nw, act_bias=ab, act=True, res1=r1, res2=r2), "plain": {}, "noise": dict(noise=nz, noise_w=nw),
          "act": dict(act_bias=ab, act=True), "res1": dict(res1=r1), "res12": dict(res1=r1, res2=r2), "noiseact": dict(noise=nz, noise_w=nw, act_bias=ab, act=True),
          "actres": dict(act_bias=ab, act=True, res1=r1)}[VAR]
    got = H.blur_fused(x, k, pad, **kw).float()
    kw2 = {a: (v.float() if torch.is_tensor(v) and v.dtype == torch.bfloat16 else v) for a, v in kw.items()}
    ref = b16(H.blur_fused(x.float(), k, pad, **kw2)).float()
    d = (got - ref).abs()
    if d.max().item() > 0 or not torch.isfinite(got).all():
        bad = (d > 0) | ~torch.isfinite(got)
        idx = bad.nonzero()
        if os.environ.get("QUIET"): nbad = globals().get("nbad", 0) + int(bad.sum()); globals()["nbad"] = nbad; continue
        print("iter", it, "n bad", int(bad.sum()), "first", idx[:6].tolist(), "last", idx[-3:].tolist(), "x ptr % 64:", x.data_ptr() % 64, "got", got[bad][:4].tolist(), "ref", ref[bad][:4].tolist())
print("done", VAR, "bad elements total", globals().get("nbad", 0))
