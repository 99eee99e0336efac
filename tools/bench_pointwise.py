"""ToRGB (modulated 1x1 conv to 3 channels + bias + the FIR-upsampled RGB skip evaluated in place) as an HBM stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
k = torch.tensor([1., 3., 3., 1.]); k = (k[:, None] * k[None, :]); k = (k / k.sum() * 4).cuda().contiguous()
def t(f, n=10):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1000
for dt in (torch.float32, torch.bfloat16):
    for (B, C, S) in [(8, 32, 1024), (8, 64, 512), (8, 128, 256), (8, 512, 64)] if dt == torch.float32 else [(16, 32, 1024), (16, 64, 512), (16, 128, 256)]:
        x = torch.randn(B, C, S, S, device="cuda").to(dt)
        w = torch.randn(3, C, device="cuda"); s_ = torch.rand(B, C, device="cuda") + 0.5; bias = torch.zeros(3, device="cuda")
        up = torch.randn(B, 3, S // 2, S // 2, device="cuda")
        us0 = t(lambda: H.pointwise(x, w, in_scale=s_, ch_bias=bias))
        us1 = t(lambda: H.pointwise(x, w, in_scale=s_, ch_bias=bias, up_src=up, up_kernel=k))
        by = B * S * S * (C * x.element_size() + 12.0)
        print(f"{str(dt)[6:]} C={C} S={S}: plain {us0:.0f} us {by/us0/1e6:.2f} TB/s | + upsampled skip {us1:.0f} us {by/us1/1e6:.2f} TB/s")
