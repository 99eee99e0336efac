"""Time ONE shape on the F(4x4,3x3) pair, kernels separately (the current VSP_CONV_DBG ablation applies).  usage: wino4_ablate.py B Cin Cout S"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
B, Cin, Cout, S = (int(v) for v in sys.argv[1:5])
x = torch.randn(B, Cin, S, S, device="cuda")
sc = torch.rand(B, Cin, device="cuda") + 0.5
w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(Cin * 9)
pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, 3, 3, 1, (1,), (1,))
f = lambda: H.conv2d_packed(x, pc, in_scale=sc, winograd=4)
f(); f(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): f()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 100
print(f"dbg={os.environ.get('VSP_CONV_DBG', '0'):>6}  {Cin}->{Cout} @{S} B{B}: {us:.0f} us  {2.0 * B * Cout * Cin * 9 * S * S / us / 1e6:.1f} eff. TF")
