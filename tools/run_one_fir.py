"""Run the 4x4 blur on one plane shape a few times (for rocprofv3 --pmc passes).  usage: run_one_fir.py B C S [bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
B, C, S = (int(v) for v in sys.argv[1:4])
bf = len(sys.argv) > 4 and sys.argv[4] == "bf16"
k = torch.tensor([1., 3., 3., 1.]); k = (k[:, None] * k[None, :]); k = (k / k.sum() * 4).cuda()
x = torch.randn(B, C, S + 1, S + 1, device="cuda")
if bf:
    x = x.to(torch.bfloat16)
nz = torch.randn(B, 1, S, S, device="cuda"); nw = torch.ones(1, device="cuda"); ab = torch.zeros(C, device="cuda")
for _ in range(3):
    H.blur_fused(x, k, (1, 1), noise=nz, noise_w=nw, act_bias=ab, act=True)
torch.cuda.synchronize()
print("done")
