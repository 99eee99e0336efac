"""Run one conv shape a few times (for rocprofv3 --pmc passes).  usage: run_one_conv.py <shape-name> <cfg-id or 0> [iters]"""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from bench_conv import SHAPES
name, cfg = sys.argv[1], sys.argv[2]
cfg = H.CONFIG_IDS[cfg] if not cfg.isdigit() else int(cfg)
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
(_, B, Cin, Cout, Hh, Ww, k, s, p, d, G) = [r for r in SHAPES if r[0] == name][0]
x = torch.randn(B, Cin, Hh, Ww, device="cuda")
if G == 1:
    w = torch.randn(Cout, Cin, k, k, device="cuda") / math.sqrt(Cin * k * k)
    pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, k, k, s, (d,), (p,))
else:
    wp = torch.randn(4, k * k, Cin, Cout // 4, device="cuda") / math.sqrt(Cin * k * k)
    pc = H.PackedConv(wp, 4, Cout // 4, Cin, k, k, 1, (1, 2, 4, 8), (1, 2, 4, 8))
oh, ow = H.conv2d_out_size(Hh, Ww, pc)
out = torch.empty(B, Cout, oh, ow, device="cuda")
for _ in range(iters):
    H.conv2d_packed(x, pc, out=out, tile_hint=cfg)
torch.cuda.synchronize()
print("done", name, cfg)
