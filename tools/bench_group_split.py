"""A SMART dilation-group launch (d = 1, 2, 4, 8 over one shared input): what the tuned table picks, the F(2x2) kernels, and the fused F(4x4)
kernel's group launch (conv_wino4f.hip: one partition of workgroups per group, dilated groups through the LDS window loader)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
def t(f, n=10):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1000
B = int(os.environ.get("B", 8))
for cin, cg, hw in [(64, 16, 512), (128, 32, 256), (256, 64, 128), (512, 128, 64), (512, 128, 32)]:
    g_ = torch.Generator().manual_seed(1)
    x = torch.randn(B, cin, hw, hw, generator=g_).cuda()
    ws = [torch.randn(cg, cin, 3, 3, generator=g_) / math.sqrt(cin * 9) for _ in range(4)]
    s_in = (torch.rand(B, cin, generator=g_) + 0.5).cuda()
    demod = (torch.rand(B, 4 * cg, generator=g_) + 0.5).cuda()
    bias = torch.randn(4 * cg, generator=g_).cuda()
    wp = torch.stack([H.pack_weight(w_.cuda())[0] for w_ in ws]).contiguous()
    pc4 = H.PackedConv(wp, 4, cg, cin, 3, 3, 1, (1, 2, 4, 8), (1, 2, 4, 8))
    out = torch.empty(B, 4 * cg, hw, hw, device="cuda")
    def one():
        return H.conv2d_packed(x, pc4, out=out, in_scale=s_in, out_scale=demod, act2=1, bias2=bias)
    def fused():
        return H.conv2d_packed(x, pc4, out=out, in_scale=s_in, out_scale=demod, act2=1, bias2=bias, winograd=5)
    def f22():
        return H.conv2d_packed(x, pc4, out=out, in_scale=s_in, out_scale=demod, act2=1, bias2=bias, winograd=True)
    y1 = f22().clone(); fused(); d = (out - y1).abs().max().item()
    us0, us1, us2 = t(one), t(f22), t(fused)
    print(f"{cin} -> 4 x {cg} at {hw}^2: table {us0:.0f} us | F(2x2) kernels {us1:.0f} us | fused F(4x4), one launch {us2:.0f} us | max diff {d:.1e}", flush=True)
