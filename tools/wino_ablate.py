"""Time ONE Winograd shape (the current VSP_CONV_DBG ablation applies).  usage: wino_ablate.py Cin Cout S [G]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
Cin, Cout, S = (int(v) for v in sys.argv[1:4])
G = int(sys.argv[4]) if len(sys.argv) > 4 else 1
B = 8
x = torch.randn(B, Cin, S, S, device="cuda")
sc = torch.rand(B, Cin, device="cuda") + 0.5
if G == 1:
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(Cin * 9)
    pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, 3, 3, 1, (1,), (1,))
else:
    wp = torch.randn(4, 9, Cin, Cout // 4, device="cuda") / math.sqrt(Cin * 9)
    pc = H.PackedConv(wp, 4, Cout // 4, Cin, 3, 3, 1, (1, 2, 4, 8), (1, 2, 4, 8))
f = lambda: H.conv2d_packed(x, pc, in_scale=sc, winograd=True, wino_form=int(os.environ.get("WINO_FORM", "0")))
f(); f(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): f()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 100
print(f"dbg={os.environ.get('VSP_CONV_DBG', '0'):>6}  {Cin}->{Cout} @{S} G={G}: {us:.0f} us  {2.0 * B * Cout * Cin * 9 * S * S / us / 1e6:.1f} eff. TF")
