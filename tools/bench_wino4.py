"""Winograd F(4x4,3x3) (vsp_conv2d_winograd4_f32: input transform + barrier-free GEMM) against F(2x2,3x3) and the direct kernel on the
deep stride-1 3x3 layers; error of each against an fp64 convolution on the first image."""
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
shapes = [(8, 512, 512, 64), (8, 256, 256, 128), (8, 512, 512, 32), (8, 256, 256, 32), (8, 256, 512, 32), (8, 128, 128, 256), (8, 128, 128, 64),
          (8, 512, 512, 16), (2, 64, 96, 48)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for (B, Cin, Cout, S) in shapes:
    torch.manual_seed(0)
    x = torch.randn(B, Cin, S, S, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(Cin * 9)
    pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, 3, 3, 1, (1,), (1,))
    sc = torch.rand(B, Cin, device="cuda") + 0.5
    nz = torch.randn(B, 1, S, S, device="cuda")
    nw = torch.tensor([0.3], device="cuda")
    res = torch.randn(B, Cout, S, S, device="cuda")
    b1 = torch.randn(Cout, device="cuda")
    dm = torch.rand(B, Cout, device="cuda") + 0.5
    kw = dict(in_scale=sc, out_scale=dm, noise=nz, noise_w=nw, bias2=b1, act2=1, res1=res)
    fl = 2.0 * B * Cout * Cin * 9 * S * S
    ref = F.conv2d((x[:1] * sc[:1, :, None, None]).double(), w.double(), padding=1) * dm[:1, :, None, None].double()
    ref = F.leaky_relu(ref + 0.3 * nz[:1].double() + b1.double()[None, :, None, None], 0.2) * math.sqrt(2) + res[:1].double()
    out = {}
    for name, wn in (("direct", False), ("F2", True), ("F4", 4)):
        y = H.conv2d_packed(x, pc, winograd=wn, **kw)
        err = (y[:1].double() - ref).abs()
        us = t(lambda: H.conv2d_packed(x, pc, winograd=wn, **kw))
        out[name] = (us, err.max().item(), err.pow(2).mean().sqrt().item())
    d = (H.conv2d_packed(x, pc, winograd=4, **kw) - H.conv2d_packed(x, pc, winograd=False, **kw)).abs().max().item()
    print(f"{Cin}->{Cout} @{S} B{B}: " + " | ".join(f"{k} {v[0]:.0f} us {fl / v[0] / 1e6:.0f} TF err max {v[1]:.1e} rms {v[2]:.1e}" for k, v in out.items()) + f" | F4 vs direct all images {d:.1e}", flush=True)
