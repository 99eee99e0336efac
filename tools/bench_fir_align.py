"""Round 3 probe: is the blur bound by its 4-byte-aligned window loads?  The same 4x4 blur on a (2S+1)-wide plane (rows start at
every dword alignment) and on a plane padded to 2S+4 columns (every row 16-byte aligned), same output size."""
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
for (B, C, S) in [(8, 32, 1024), (8, 64, 512), (8, 128, 256)]:
    x = torch.randn(B, C, S + 1, S + 1, device="cuda")
    xa = torch.randn(B, C, S + 1, S + 4, device="cuda")
    us0 = t(lambda: H.blur_fused(x, k, (1, 1)))
    us1 = t(lambda: H.blur_fused(xa, k, (1, 1, 1, -2)) if False else H.upfirdn2d_native_layout(xa.view(-1, S + 1, S + 4, 1), k, 1, 1, 1, 1, 1, -2, 1, 1))
    by = B * C * S * S * 8.0
    print(f"C={C} S={S}: odd-width rows {us0:.0f} us {by/us0/1e6:.2f} TB/s | 16-byte aligned rows {us1:.0f} us {by/us1/1e6:.2f} TB/s")
    y = torch.empty(B * C * S * S, device="cuda"); z = torch.randn(B * C * S * S, device="cuda")
    usc = t(lambda: y.copy_(z))
    print(f"    plain copy of the same bytes: {usc:.0f} us {by/usc/1e6:.2f} TB/s")
