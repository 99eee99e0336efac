"""Run one Winograd conv launch a few times (for rocprofv3 --pmc passes).  usage: run_one_wino.py Cin Cout S [G] [iters]
env WINO = 4: the F(4x4,3x3) pair (vsp_conv2d_winograd4_f32) instead of F(2x2,3x3); WINO = 5: the fused F(4x4,3x3) kernel (vsp_conv2d_winograd4f_f32).
G = 1: ordinary 3x3 layer Cin -> Cout; G = 4: the four dilation groups (1, 2, 4, 8) of a SMART layer, Cout / 4 channels each."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
Cin, Cout, S = (int(v) for v in sys.argv[1:4])
G = int(sys.argv[4]) if len(sys.argv) > 4 else 1
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
B = 8
x = torch.randn(B, Cin, S, S, device="cuda")
sc = torch.rand(B, Cin, device="cuda") + 0.5
if G == 1:
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(Cin * 9)
    pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, 3, 3, 1, (1,), (1,))
else:
    wp = torch.randn(4, 9, Cin, Cout // 4, device="cuda") / math.sqrt(Cin * 9)
    pc = H.PackedConv(wp, 4, Cout // 4, Cin, 3, 3, 1, (1, 2, 4, 8), (1, 2, 4, 8))
for _ in range(iters):
    H.conv2d_packed(x, pc, in_scale=sc, winograd={"4": 4, "5": 5}.get(os.environ.get("WINO"), True), wino_form=int(os.environ.get("WINO_FORM", "0")))
torch.cuda.synchronize()
print("done", Cin, Cout, S, G)
