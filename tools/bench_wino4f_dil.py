"""Fused F(4x4) kernel on DILATED one-group layers (LDS window loader, conv_wino4f.hip D = 2 / 4) and, with VSP_WINO4F_LDS=1, the same loader
at dilation 1: error against float64 F.conv2d on image 0 and time against the F(2x2) kernels on the same one-group layer."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from vspbfr_amd import hip_ops as H
def t(f, n=10):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1000
B = int(os.environ.get("B", 8))
shapes = [(64, 64, 512), (128, 32, 256), (256, 64, 128), (64, 16, 512), (32, 32, 64)]
if os.environ.get("QUICK"): shapes = [(32, 32, 64), (128, 32, 256)]
for cin, cout, hw in shapes:
    for d in (1, 2, 4, 8):
        g_ = torch.Generator().manual_seed(d)
        x = torch.randn(B, cin, hw, hw, generator=g_).cuda()
        w = torch.randn(cout, cin, 3, 3, generator=g_) / math.sqrt(cin * 9)
        s_in = (torch.rand(B, cin, generator=g_) + 0.5).cuda()
        demod = (torch.rand(B, cout, generator=g_) + 0.5).cuda()
        bias = torch.randn(cout, generator=g_).cuda()
        res = torch.randn(B, cout, hw, hw, generator=g_).cuda()
        nz = torch.randn(B, 1, hw, hw, generator=g_).cuda()
        nw = torch.tensor([0.3], device="cuda")
        pc = H.PackedConv(H.pack_weight(w.cuda()), 1, cout, cin, 3, 3, 1, (d,), (d,))
        out = torch.empty(B, cout, hw, hw, device="cuda")
        kw = dict(out=out, in_scale=s_in, out_scale=demod, act2=1, bias2=bias, res1=res, noise=nz, noise_w=nw)
        H.conv2d_packed(x, pc, winograd=5, **kw)
        y = out.clone()
        xr = (x[:1].double().cpu() * s_in[:1].double().cpu()[:, :, None, None])
        ref = F.conv2d(xr, w.double(), padding=d, dilation=d) * demod[:1].double().cpu()[:, :, None, None]
        ref = ref + 0.3 * nz[:1].double().cpu() + bias.double().cpu()[None, :, None, None]
        ref = F.leaky_relu(ref, 0.2) * math.sqrt(2.0) + res[:1].double().cpu()
        err = (y[:1].double().cpu() - ref).abs().max().item()
        H.conv2d_packed(x, pc, winograd=True, **kw)
        dk = (out - y).abs().max().item()
        kp = dict(out=out, in_scale=s_in, out_scale=demod)
        u5 = t(lambda: H.conv2d_packed(x, pc, winograd=5, **kp))
        u2 = t(lambda: H.conv2d_packed(x, pc, winograd=True, **kp))
        fl = 2.0 * B * cout * cin * 9 * hw * hw
        print(f"{cin} -> {cout} at {hw}^2 d = {d} B = {B}: fused F(4x4) {u5:.0f} us ({fl / u5 / 1e6:.0f} TF) | F(2x2) {u2:.0f} us | err vs float64 {err:.1e} | all images vs F(2x2) {dk:.1e}", flush=True)
