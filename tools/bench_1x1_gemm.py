"""1x1 convolutions on small maps (identity-network bottlenecks): tiled conv kernel vs vsp_conv1x1_small_f32 vs the library GEMM,
device time by events.  usage: python tools/bench_1x1_gemm.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H


def t(fn, n=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (cin, cout, hw, B) in [(1024, 256, 7, 8), (256, 1024, 7, 8), (2048, 512, 4, 8), (512, 2048, 4, 8), (512, 128, 14, 8), (128, 512, 14, 8),
                           (256, 64, 28, 8), (64, 256, 28, 8), (1024, 2048, 4, 8), (512, 1024, 7, 8), (256, 512, 14, 8)]:
    x = torch.randn(B, cin, hw, hw, device="cuda")
    w = torch.randn(cout, cin, 1, 1, device="cuda") * 0.03
    pc = H.PackedConv(H.pack_weight(w), 1, cout, cin, 1, 1, 1, (1,), (0,))
    w2 = w.view(cout, cin)
    print((cin, cout, hw), "tiled conv %.1f us   small-map gemm %.1f us   library matmul %.1f us" % (
        t(lambda: H.conv2d_packed(x, pc)), t(lambda: H.conv1x1_small(x, w2)), t(lambda: torch.matmul(w2, x.view(B, cin, hw * hw)))))
