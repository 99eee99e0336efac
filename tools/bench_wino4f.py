"""Fused Winograd F(4x4,3x3) (vsp_conv2d_winograd4f_f32, round 5: transform in registers, no V image) against the F(4x4) pair, F(2x2,3x3) and
the direct kernel on the shallow wide stride-1 3x3 layers; error of each against an fp64 convolution on the first image and the fused form
against the direct kernel on ALL images.  args: B,Cin,Cout,H[,W] ...; env FORMS=F4f,F2 picks the kernels."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from vspbfr_amd import hip_ops as H
def t(f, n=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1000
shapes = [(2, 64, 64, 64), (2, 8, 16, 20, 36), (3, 40, 48, 16, 80), (1, 256, 96, 8, 16), (8, 64, 64, 512), (8, 32, 32, 1024), (8, 64, 64, 256), (8, 64, 64, 128), (8, 128, 128, 256), (8, 64, 128, 128)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
forms = os.environ.get("FORMS", "direct,F2,F4,F4f").split(",")
for shp in shapes:
    B, Cin, Cout, S = shp[:4]
    Wd = shp[4] if len(shp) > 4 else S
    torch.manual_seed(0)
    x = torch.randn(B, Cin, S, Wd, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(Cin * 9)
    pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, 3, 3, 1, (1,), (1,))
    sc = torch.rand(B, Cin, device="cuda") + 0.5
    nz = torch.randn(B, 1, S, Wd, device="cuda")
    nw = torch.tensor([0.3], device="cuda")
    res = torch.randn(B, Cout, S, Wd, device="cuda")
    b1 = torch.randn(Cout, device="cuda")
    dm = torch.rand(B, Cout, device="cuda") + 0.5
    nores = bool(os.environ.get("NORES"))          # NORES=1: the StyledConv operand set (no residual)
    kw = dict(in_scale=sc, out_scale=dm, noise=nz, noise_w=nw, bias2=b1, act2=1)
    if not nores:
        kw["res1"] = res
    fl = 2.0 * B * Cout * Cin * 9 * S * Wd
    ref = F.conv2d((x[:1] * sc[:1, :, None, None]).double(), w.double(), padding=1) * dm[:1, :, None, None].double()
    ref = F.leaky_relu(ref + 0.3 * nz[:1].double() + b1.double()[None, :, None, None], 0.2) * math.sqrt(2) + (0 if nores else res[:1].double())
    out = {}
    for name, wn in (("direct", False), ("F2", True), ("F4", 4), ("F4f", 5)):
        if name not in forms:
            continue
        if wn == 5 and not H.winograd4f_eligible(pc, S, Wd, S, Wd):
            continue
        y = H.conv2d_packed(x, pc, winograd=wn, **kw)
        err = (y[:1].double() - ref).abs()
        us = t(lambda: H.conv2d_packed(x, pc, winograd=wn, **kw))
        out[name] = (us, err.max().item(), err.pow(2).mean().sqrt().item())
    d = (H.conv2d_packed(x, pc, winograd=5, **kw) - H.conv2d_packed(x, pc, winograd=False, **kw)).abs().max().item() if "F4f" in out else float("nan")
    print(f"{Cin}->{Cout} @{S}x{Wd} B{B}: " + " | ".join(f"{k} {v[0]:.0f} us {fl / v[0] / 1e6:.0f} TF err max {v[1]:.1e} rms {v[2]:.1e}" for k, v in out.items()) + f" | F4f vs direct all images {d:.1e}", flush=True)
