"""Time one layer on vsp_conv2d_bf16rv (the current VSP_CONV_DBG ablation of a tools/build_abl.sh build applies).
usage: bf16rv_ablate.py B Cin Cout S [hint]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
B, Cin, Cout, S = (int(v) for v in sys.argv[1:5])
hint = int(sys.argv[5]) if len(sys.argv) > 5 else 0
x = torch.randn(B, Cin, S, S, device="cuda").to(torch.bfloat16)
w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(Cin * 9)
pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, 3, 3, 1, (1,), (1,))
sc = torch.rand(B, Cin, device="cuda") + 0.5
nz = torch.randn(B, 1, S, S, device="cuda"); nw = torch.tensor([0.3], device="cuda"); b2 = torch.randn(Cout, device="cuda")
out = torch.empty(B, Cout, S, S, device="cuda", dtype=torch.bfloat16)
kind = os.environ.get("KIND", "rv")
f = lambda: H.conv2d_packed(x, pc, in_scale=sc, noise=nz, noise_w=nw, bias2=b2, act2=1, bf16=("rv" if kind == "rv" else True), tile_hint=hint, out=out)
f(); f(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): f()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 100
by = B * S * S * (Cin + Cout) * 2.0
print(f"{kind} dbg={os.environ.get('VSP_CONV_DBG', '0'):>4}  {Cin}->{Cout} @{S} B{B}: {us:.0f} us  {2.0 * B * Cout * Cin * 9 * S * S / us / 1e6:.0f} TF  {by / us / 1e6:.2f} TB/s algorithmic", flush=True)
