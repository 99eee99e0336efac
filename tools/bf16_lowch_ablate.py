"""Time one stride-1 3x3 layer on the bf16 kernel with bf16 activations in HBM (the current VSP_CONV_DBG ablation applies).
usage: bf16_lowch_ablate.py B Cin Cout S"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
B, Cin, Cout, S = (int(v) for v in sys.argv[1:5])
H.BF16_CONV = True
H.ACT_BF16 = True
x = torch.randn(B, Cin, S, S, device="cuda").to(torch.bfloat16)
w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(Cin * 9)
pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, 3, 3, 1, (1,), (1,))
sc = torch.rand(B, Cin, device="cuda") + 0.5
nz = torch.randn(B, 1, S, S, device="cuda"); nw = torch.tensor([0.3], device="cuda"); b2 = torch.randn(Cout, device="cuda")
f = lambda: H.conv2d_packed(x, pc, in_scale=sc, noise=nz, noise_w=nw, bias2=b2, act2=1, bf16=True)
y = f(); assert y.dtype == torch.bfloat16
f(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): f()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) * 100
by = B * S * S * (Cin + Cout) * 2.0
print(f"dbg={os.environ.get('VSP_CONV_DBG', '0'):>8}  {Cin}->{Cout} @{S} B{B}: {us:.0f} us  {2.0 * B * Cout * Cin * 9 * S * S / us / 1e6:.0f} TF  {by / us / 1e6:.2f} TB/s algorithmic")
