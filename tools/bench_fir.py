"""Time the Blur FIR (4x4 taps, pad (1,1)) after the up-convs of the path, with and without the fused epilogue."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
k = torch.tensor([1., 3., 3., 1.]); k = (k[:, None] * k[None, :]); k = (k / k.sum() * 4).cuda()
def t(f, n=10):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1000
for (B, C, S) in [(8, 32, 1024), (8, 64, 512), (8, 128, 256), (8, 256, 128), (8, 512, 64)]:
    x = torch.randn(B, C, S + 1, S + 1, device="cuda")
    nz = torch.randn(B, 1, S, S, device="cuda"); nw = torch.ones(1, device="cuda"); ab = torch.zeros(C, device="cuda")
    r1 = torch.randn(B, C, S, S, device="cuda"); r2 = torch.randn(B, C, S, S, device="cuda")
    us0 = t(lambda: H.blur_fused(x, k, (1, 1)))
    us1 = t(lambda: H.blur_fused(x, k, (1, 1), noise=nz, noise_w=nw, act_bias=ab, act=True))
    us2 = t(lambda: H.blur_fused(x, k, (1, 1), noise=nz, noise_w=nw, act_bias=ab, act=True, res1=r1, res2=r2))
    by = B * C * S * S * 8.0
    print(f"C={C} S={S}: plain {us0:.0f} us {by/us0/1e6:.2f} TB/s | noise+act {us1:.0f} us {by/us1/1e6:.2f} TB/s | +2 res {us2:.0f} us {by*2/us2/1e6:.2f} TB/s")
# bf16 tensors (BASELINE configs[2]: bf16 activations in HBM): 2 B read + 2 B written per output element
for (B, C, S) in [(16, 32, 1024), (16, 64, 512), (16, 128, 256)]:
    x = torch.randn(B, C, S + 1, S + 1, device="cuda").to(torch.bfloat16)
    nz = torch.randn(B, 1, S, S, device="cuda"); nw = torch.ones(1, device="cuda"); ab = torch.zeros(C, device="cuda")
    us0 = t(lambda: H.blur_fused(x, k, (1, 1)))
    us1 = t(lambda: H.blur_fused(x, k, (1, 1), noise=nz, noise_w=nw, act_bias=ab, act=True))
    by = B * C * S * S * 4.0
    print(f"bf16 C={C} S={S}: plain {us0:.0f} us {by/us0/1e6:.2f} TB/s | noise+act {us1:.0f} us {by/us1/1e6:.2f} TB/s")
