"""vsp_conv2d_bf16rv (row-vector K) against vsp_conv2d_bf16 on the low-channel large-map stride-1 layers of the bf16-activation
configuration: time per launch (HIP events, StyledConv epilogue: style scale, demodulation, noise, bias, leaky ReLU), algorithmic
bytes per second, and the error of both against the fp32 direct kernel.
usage: bench_bf16rv.py [B,Cin,Cout,H ...]   (default: the configs[2] layers at batch 16)"""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
dev = torch.device("cuda", 0)
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:] if not a.startswith("g")] or [
    (16, 64, 64, 512), (16, 32, 32, 1024), (16, 128, 128, 256), (16, 64, 64, 256), (16, 64, 64, 128), (16, 128, 128, 64), (16, 256, 256, 128), (16, 64, 128, 128)]
hints = [int(v) for v in os.environ.get("HINTS", "0").split(",")]


def timed(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


groups = [tuple(int(v) for v in a[1:].split(",")) for a in sys.argv[1:] if a.startswith("g")]
shapes = [t for t in shapes if len(t) == 4]
for B, Cin, Cg, S in groups or ([] if len(sys.argv) > 1 else [(16, 64, 16, 512), (16, 128, 32, 256), (16, 256, 64, 128)]):
    torch.manual_seed(0)
    x = torch.randn(B, Cin, S, S, device=dev).to(torch.bfloat16)
    wp = torch.stack([H.pack_weight(torch.randn(Cg, Cin, 3, 3, device=dev) / math.sqrt(Cin * 9))[0] for _ in range(4)]).contiguous()
    pc = H.PackedConv(wp, 4, Cg, Cin, 3, 3, 1, (1, 2, 4, 8), (1, 2, 4, 8))
    kw = dict(in_scale=torch.rand(B, Cin, device=dev) + 0.5)
    out = torch.empty(B, 4 * Cg, S, S, device=dev, dtype=torch.bfloat16)
    H.BF16_RV = False
    H.BF16_DG = False
    t0 = timed(lambda: H.conv2d_packed(x, pc, bf16=True, out=out, **kw))
    y0 = out.float()
    t1 = timed(lambda: H.conv2d_packed(x, pc, bf16="rv", out=out, **kw))
    y1 = out.float()
    nb = (x.numel() + out.numel()) * 2
    fl = 2.0 * B * 4 * Cg * Cin * 9 * S * S
    if H.bf16dg_eligible(pc, S, S, S, S):   # round 5: the dilation-group kernel (conv_bf16_dg.hip)
        t2 = timed(lambda: H.conv2d_packed(x, pc, bf16="dg", out=out, **kw))
        y2 = out.float()
        print(f"{Cin}->4x{Cg} dilated @{S} B{B}: dg {t2:.0f} us {nb / t2 / 1e6:.2f} TB/s {fl / t2 / 1e6:.0f} TF x{t0 / t2:.2f} vs bf16, maxdiff vs bf16 {float((y2 - y0).abs().max()):.3g}", flush=True)
    print(f"{Cin}->4x{Cg} dilated @{S} B{B}: bf16 {t0:.0f} us {nb / t0 / 1e6:.2f} TB/s {fl / t0 / 1e6:.0f} TF | rv {t1:.0f} us {nb / t1 / 1e6:.2f} TB/s {fl / t1 / 1e6:.0f} TF x{t0 / t1:.2f} maxdiff vs bf16 {float((y1 - y0).abs().max()):.3g} (range {float(y0.abs().max()):.2f})", flush=True)

for B, Cin, Cout, S in shapes:
    torch.manual_seed(0)
    x = torch.randn(B, Cin, S, S, device=dev).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, 3, 3, device=dev) / math.sqrt(Cin * 9)
    pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, 3, 3, 1, (1,), (1,))
    kw = dict(in_scale=torch.rand(B, Cin, device=dev) + 0.5, out_scale=torch.rand(B, Cout, device=dev) + 0.5, act2=1,
              bias2=torch.randn(Cout, device=dev), noise=torch.randn(B, 1, S, S, device=dev), noise_w=torch.full((1,), 0.2, device=dev))
    out = torch.empty(B, Cout, S, S, device=dev, dtype=torch.bfloat16)
    t0 = timed(lambda: H.conv2d_packed(x, pc, bf16=True, out=out, **kw))
    y0 = out.float()
    nb = (x.numel() + out.numel()) * 2 + B * S * S * 4
    fl = 2.0 * B * Cout * Cin * 9 * S * S
    line = f"{Cin}->{Cout} @{S} B{B}: bf16 {t0:.0f} us {nb / t0 / 1e6:.2f} TB/s {fl / t0 / 1e6:.0f} TF"
    for h in hints:
        if not H.bf16rv_eligible(pc, S, S, S, S):
            line += " | rv: not eligible"
            y1 = y0
            break
        t1 = timed(lambda: H.conv2d_packed(x, pc, bf16="rv", tile_hint=h, out=out, **kw))
        y1 = out.float()
        line += f" | rv[{h}] {t1:.0f} us {nb / t1 / 1e6:.2f} TB/s {fl / t1 / 1e6:.0f} TF x{t0 / t1:.2f} maxdiff vs bf16 {float((y1 - y0).abs().max()):.3g}"
    if B * Cout * S * S <= 2 ** 28:
        ref = H.conv2d_packed(x.float(), pc, bf16=False, winograd=False, **kw)
        line += f" | err vs fp32: bf16 {float((y0 - ref).abs().max()):.3g} rv {float((y1 - ref).abs().max()):.3g} (range {float(ref.abs().max()):.2f})"
    print(line, flush=True)
