"""Micro-benchmark of vsp_conv2d_f32 on the path's dominant layer shapes (GPU box).  Prints TFLOP/s per tile config."""
import math
import sys
import os
import json

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vspbfr_amd import hip_ops as H
from vspbfr_amd._lib import lib

SHAPES = [
    # name, B, Cin, Cout, H, W, k, stride, pad, dil, groups(dilated)
    ("fusion512", 8, 64, 64, 512, 512, 3, 1, 1, 1, 1),
    ("fusion256", 8, 128, 128, 256, 256, 3, 1, 1, 1, 1),
    ("fusion128", 8, 256, 256, 128, 128, 3, 1, 1, 1, 1),
    ("fusion64", 8, 512, 512, 64, 64, 3, 1, 1, 1, 1),
    ("fusion32", 8, 512, 512, 32, 32, 3, 1, 1, 1, 1),
    ("dil512", 8, 64, 64, 512, 512, 3, 1, 0, 0, 4),
    ("dil256", 8, 128, 128, 256, 256, 3, 1, 0, 0, 4),
    ("dil64", 8, 512, 512, 64, 64, 3, 1, 0, 0, 4),
    ("sg1024", 4, 32, 32, 1024, 1024, 3, 1, 1, 1, 1),
    ("down256", 8, 128, 256, 257, 257, 3, 2, 0, 1, 1),
    ("head64", 8, 512, 512, 64, 64, 3, 2, 1, 1, 1),
    ("irse128", 8, 64, 64, 128, 128, 3, 1, 1, 1, 1),
    ("irse32", 8, 256, 256, 32, 32, 3, 1, 1, 1, 1),
    ("c16", 8, 512, 512, 16, 16, 3, 1, 1, 1, 1),
    ("c8", 8, 512, 512, 8, 8, 3, 1, 1, 1, 1),
    ("c4", 8, 512, 512, 4, 4, 3, 1, 1, 1, 1),
]


def bench(fn, iters=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    only = sys.argv[1].split(",") if len(sys.argv) > 1 else None
    n = lib.vsp_conv2d_num_configs()
    res = {}
    for (name, B, Cin, Cout, Hh, Ww, k, s, p, d, G) in SHAPES:
        if only and name not in only:
            continue
        x = torch.randn(B, Cin, Hh, Ww, device="cuda")
        if G == 1:
            w = torch.randn(Cout, Cin, k, k, device="cuda") / math.sqrt(Cin * k * k)
            pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, k, k, s, (d,), (p,))
        else:
            wp = torch.randn(4, k * k, Cin, Cout // 4, device="cuda") / math.sqrt(Cin * k * k)
            pc = H.PackedConv(wp, 4, Cout // 4, Cin, k, k, 1, (1, 2, 4, 8), (1, 2, 4, 8))
        oh, ow = H.conv2d_out_size(Hh, Ww, pc)
        out = torch.empty(B, Cout, oh, ow, device="cuda")
        flops = 2.0 * B * Cout * oh * ow * Cin * k * k
        row = {}
        for c in range(0, n + 1):
            try:
                ms = bench(lambda: H.conv2d_packed(x, pc, out=out, tile_hint=c))
            except RuntimeError as ex:
                continue
            row["auto" if c == 0 else lib.vsp_conv2d_config_name(c - 1).decode()] = round(flops / ms / 1e9, 1)
        res[name] = row
        print(name, f"{flops/1e9:.1f} GF", row, flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/bench_conv.json", "w"), indent=1)


if __name__ == "__main__":
    main()
