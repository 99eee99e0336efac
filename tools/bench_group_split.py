"""Round 5 experiment: a SMART dilation-group launch (d = 1, 2, 4, 8 over one shared input) as ONE launch against group 0 (d = 1, a plain
convolution) on the fused F(4x4) kernel + the three dilated groups on the F(2x2) kernel."""
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
B = 8
for cin, cg, hw in [(64, 16, 512), (128, 32, 256), (256, 64, 128), (512, 128, 64), (512, 128, 32)]:
    g_ = torch.Generator().manual_seed(1)
    x = torch.randn(B, cin, hw, hw, generator=g_).cuda()
    ws = [torch.randn(cg, cin, 3, 3, generator=g_) / math.sqrt(cin * 9) for _ in range(4)]
    s_in = (torch.rand(B, cin, generator=g_) + 0.5).cuda()
    demod = (torch.rand(B, 4 * cg, generator=g_) + 0.5).cuda()
    bias = torch.randn(4 * cg, generator=g_).cuda()
    wp = torch.stack([H.pack_weight(w_.cuda())[0] for w_ in ws]).contiguous()
    pc4 = H.PackedConv(wp, 4, cg, cin, 3, 3, 1, (1, 2, 4, 8), (1, 2, 4, 8))
    pc0 = H.PackedConv(wp[:1].contiguous(), 1, cg, cin, 3, 3, 1, (1,), (1,))
    pc3 = H.PackedConv(wp[1:].contiguous(), 3, cg, cin, 3, 3, 1, (2, 4, 8), (2, 4, 8))
    out = torch.empty(B, 4 * cg, hw, hw, device="cuda")
    def one():
        return H.conv2d_packed(x, pc4, out=out, in_scale=s_in, out_scale=demod, act2=1, bias2=bias)
    w0 = 5 if cin <= 256 else 4       # (the fused kernel serves up to 256 input channels: the F(4x4) pair beyond)
    def split():
        H.conv2d_packed(x, pc0, out=out, y_coff=0, in_scale=s_in, out_scale=demod[:, :cg].contiguous(), act2=1, bias2=bias[:cg].contiguous(), winograd=w0)
        H.conv2d_packed(x, pc3, out=out, y_coff=cg, in_scale=s_in, out_scale=demod[:, cg:].contiguous(), act2=1, bias2=bias[cg:].contiguous(), winograd=True)
    y1 = one().clone(); split(); d = (out - y1).abs().max().item()
    us1, us2 = t(one), t(split)
    us0 = t(lambda: H.conv2d_packed(x, pc0, out=out, y_coff=0, in_scale=s_in, winograd=w0))
    us3 = t(lambda: H.conv2d_packed(x, pc3, out=out, y_coff=cg, in_scale=s_in, winograd=True))
    print(f"{cin} -> 4 x {cg} at {hw}^2: one launch {us1:.0f} us | split {us2:.0f} us (F4f group 0: {us0:.0f}, three dilated groups: {us3:.0f}) | max diff {d:.1e}", flush=True)
