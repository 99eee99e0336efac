"""Run one bf16 conv shape a few times (for rocprofv3 --pmc passes).  usage: run_one_bf16.py Cin Cout S variant [G] [iters]
env: IO_BF16=1 bf16 activations in HBM; RV=1 the row-vector-K kernel (vsp_conv2d_bf16rv); B=<batch> (default 8)"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
Cin, Cout, S, v = (int(a) for a in sys.argv[1:5])
G = int(sys.argv[5]) if len(sys.argv) > 5 else 1
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 3
B = int(os.environ.get("B", "8"))
x = torch.randn(B, Cin, S, S, device="cuda")
if os.environ.get("IO_BF16"):
    x = x.to(torch.bfloat16)
if G == 1:
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(Cin * 9)
    pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, 3, 3, 1, (1,), (1,))
else:
    wp = torch.randn(4, 9, Cin, Cout // 4, device="cuda") / math.sqrt(Cin * 9)
    pc = H.PackedConv(wp, 4, Cout // 4, Cin, 3, 3, 1, (1, 2, 4, 8), (1, 2, 4, 8))
sc = torch.rand(B, Cin, device="cuda") + 0.5
out = torch.empty(B, Cout, S, S, device="cuda", dtype=x.dtype)
for _ in range(iters):
    H.conv2d_packed(x, pc, out=out, in_scale=sc, bf16=("rv" if os.environ.get("RV") else True), tile_hint=v)
torch.cuda.synchronize()
print("done")
