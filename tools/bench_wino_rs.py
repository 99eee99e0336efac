"""Register-resident-U Winograd kernel (conv_wino_rs.hip, wino_form = 3) against the other forms of the same layer and against fp64 F.conv2d.

    python tools/bench_wino_rs.py [--check] [--batch 8]

Prints per layer: microseconds per launch of every form that serves it (HIP events, 20 launches after 3 warm-ups), effective TFLOP/s and,
with --check, max |delta| against float64 F.conv2d on the first image."""
import argparse
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from vspbfr_amd import hip_ops as H

LAYERS = [  # Cin, channels per group, H = W, dilations
    (64, 16, 512, (1, 2, 4, 8)),
    (64, 64, 512, (1,)),
    (32, 32, 1024, (1,)),
    (64, 64, 256, (1,)),
    (64, 64, 128, (1,)),
    (64, 16, 256, (1, 2, 4, 8)),
    (128, 32, 256, (1, 2, 4, 8)),
    (256, 64, 128, (1, 2, 4, 8)),
    (512, 128, 64, (1, 2, 4, 8)),
    (512, 128, 32, (1, 2, 4, 8)),
    (512, 128, 16, (1, 2, 4, 8)),
]


def time_us(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--only", type=str, default="")
    args = ap.parse_args()
    dev = torch.device("cuda")
    B = args.batch
    for li, (cin, cg, hw, dils) in enumerate(LAYERS):
        if args.only and str(li) not in args.only.split(","):
            continue
        G = len(dils)
        g_ = torch.Generator().manual_seed(li)
        x = torch.randn(B, cin, hw, hw, generator=g_).to(dev)
        ws = [torch.randn(cg, cin, 3, 3, generator=g_) / math.sqrt(cin * 9) for _ in dils]
        s_in = (torch.rand(B, cin, generator=g_) + 0.5).to(dev)
        demod = (torch.rand(B, G * cg, generator=g_) + 0.5).to(dev)
        bias = torch.randn(G * cg, generator=g_).to(dev)
        nz, nw = torch.randn(B, 1, hw, hw, generator=g_).to(dev), torch.tensor([0.7]).to(dev)
        wp = torch.stack([H.pack_weight(w_.to(dev))[0] for w_ in ws]).contiguous()
        pc = H.PackedConv(wp, G, cg, cin, 3, 3, 1, dils, dils)
        kw = dict(in_scale=s_in, out_scale=demod, noise=nz, noise_w=nw, act2=1, bias2=bias)
        flops = 2.0 * B * G * cg * hw * hw * cin * 9
        out = {}
        res = []
        for name, call in (("auto", dict(winograd=None)), ("direct", dict(winograd=False)), ("wino task", dict(winograd=True, wino_form=1)),
                           ("wino row-owner", dict(winograd=True, wino_form=2)), ("wino reg-U", dict(winograd=True, wino_form=3))):
            try:
                y = H.conv2d_packed(x, pc, **kw, **call)
                us = time_us(lambda: H.conv2d_packed(x, pc, **kw, **call))
            except RuntimeError as e:
                res.append(f"{name}: n/a ({str(e)[:60]})")
                continue
            out[name] = y
            res.append(f"{name}: {us:8.1f} us {flops / us / 1e6:6.1f} TF")
        print(f"[{li}] {cin} -> {G} x {cg} at {hw}^2 d={dils} B={B} | " + " | ".join(res), flush=True)
        if args.check:
            xd = (x[:1] * s_in[:1].view(1, cin, 1, 1)).double().cpu()
            ref = torch.cat([F.conv2d(xd, w_.double(), padding=d, dilation=d) for w_, d in zip(ws, dils)], dim=1)
            ref = ref * demod[:1].double().cpu().view(1, -1, 1, 1) + nz[:1].double().cpu() * 0.7
            ref = F.leaky_relu(ref + bias.double().cpu().view(1, -1, 1, 1), 0.2) * math.sqrt(2)
            for name, y in out.items():
                print(f"      {name}: max |delta| vs fp64 {float((y[:1].double().cpu() - ref).abs().max()):.3e}", flush=True)


if __name__ == "__main__":
    main()
