"""Weight-gradient kernel on the layer shapes of the training step (B = 4).  usage: python tools/bench_wgrad.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
dev = torch.device("cuda", 0)
B = int(os.environ.get("B", 4))
SHAPES = [  # cin, cout, size, stride, groups(shared-input dilation groups)
    (64, 64, 512, 1, 1), (128, 128, 256, 1, 1), (256, 256, 128, 1, 1), (512, 512, 64, 1, 1), (512, 512, 32, 1, 1), (512, 512, 16, 1, 1),
    (64, 64, 512, 1, 4), (128, 128, 256, 1, 4), (512, 512, 64, 1, 4),
    (64, 128, 513, 2, 1), (256, 512, 129, 2, 1), (512, 512, 65, 2, 1), (512, 512, 33, 2, 1),
]
for cin, cout, size, stride, G in SHAPES:
    x = torch.randn(B, cin, size, size, device=dev)
    if G == 1:
        oh = (size + 2 - 3) // stride + 1 if stride == 1 else (size - 3) // 2 + 1
        dy = torch.randn(B, cout, oh, oh, device=dev)
        args = dict(weight_shape=(cout, cin, 3, 3), stride=stride, padding=1 if stride == 1 else 0, dilation=1, groups=1)
    else:
        oh = size
        dy = torch.randn(B, cout, oh, oh, device=dev)
        args = dict(weight_shape=(cout, cin, 3, 3), stride=1, padding=(1, 2, 4, 8), dilation=(1, 2, 4, 8), groups=4, x_shared=True)
    for _ in range(2):
        H.conv2d_wgrad(x, dy, **args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 5
    for _ in range(n):
        H.conv2d_wgrad(x, dy, **args)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * B * cout * oh * oh * (cin if G == 1 else cin) * 9 / (1 if G == 1 else 1) * (1.0 if G == 1 else 1.0 / 1)
    if G > 1:
        fl = 2.0 * B * cout * oh * oh * cin * 9
    print(f"{cin}->{cout} @{size} s{stride} G{G}: {ms*1e3:.0f} us  {fl/ms/1e9:.1f} TF")
for cin, cout, size in [(3, 16, 512), (3, 64, 512)]:   # FromRGB: the stream form
    x, dy = torch.randn(B, cin, size, size, device=dev), torch.randn(B, cout, size, size, device=dev)
    for _ in range(2):
        H.conv2d_wgrad(x, dy, (cout, cin, 1, 1), 1, 0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        H.conv2d_wgrad(x, dy, (cout, cin, 1, 1), 1, 0)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"{cin}->{cout} 1x1 @{size}: {ms*1e3:.0f} us  {(x.numel() + dy.numel()) * 4 / ms / 1e6:.0f} GB/s")
