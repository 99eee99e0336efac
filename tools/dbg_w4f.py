import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
torch.manual_seed(0)
B, Cin, Cout, S, Wd = 2, 16, 32, 8, 64
x = torch.randn(B, Cin, S, Wd, device="cuda")
w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(Cin * 9)
pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, 3, 3, 1, (1,), (1,))
sc = torch.rand(B, Cin, device="cuda") + 0.5
nz = torch.randn(B, 1, S, Wd, device="cuda")
nw = torch.tensor([0.3], device="cuda")
res = torch.randn(B, Cout, S, Wd, device="cuda")
b1 = torch.randn(Cout, device="cuda")
dm = torch.rand(B, Cout, device="cuda") + 0.5
cases = {"plain": {}, "in_scale": dict(in_scale=sc), "out_scale": dict(out_scale=dm), "noise": dict(noise=nz, noise_w=nw), "bias2": dict(bias2=b1),
         "act2": dict(bias2=b1, act2=1), "res1": dict(res1=res), "act1": dict(bias1=b1, act1=True), "all": dict(in_scale=sc, out_scale=dm, noise=nz, noise_w=nw, bias2=b1, act2=1, res1=res)}
for name, kw in cases.items():
    y5 = H.conv2d_packed(x, pc, winograd=5, **kw)
    y0 = H.conv2d_packed(x, pc, winograd=False, **kw)
    print(name, "max diff %.3g" % (y5 - y0).abs().max().item(), "per image", (y5 - y0).abs().amax(dim=(1, 2, 3)).tolist())
