"""Run one layer of tools/bench_pipe.py's shape list on a named tile configuration a few times (rocprofv3 --pmc passes).
usage: run_one_pipe.py <shape> <config name | tuned> [iters]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from vspbfr_amd import hip_ops as H
from bench_pipe import SHAPES
name, cfg = sys.argv[1], sys.argv[2]
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
B, Cin, Cout, Hh, Ww, s, pad, G, tr = SHAPES[name]
x = torch.randn(B, Cin, Hh, Ww, device="cuda")
sc = torch.rand(B, Cin, device="cuda") + 0.5
if G == 1:
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(Cin * 9)
    pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, 3, 3, s, (1,), (pad,))
else:
    wp = torch.randn(4, 9, Cin, Cout // 4, device="cuda") / math.sqrt(Cin * 9)
    pc = H.PackedConv(wp, 4, Cout // 4, Cin, 3, 3, 1, (1, 2, 4, 8), (1, 2, 4, 8))
kw = dict(in_scale=sc, transposed=tr, winograd=False, bf16=False)
if cfg != "tuned":
    kw["tile_hint"] = H.CONFIG_IDS[cfg]
for _ in range(iters):
    H.conv2d_packed(x, pc, **kw)
torch.cuda.synchronize()
print("done", name, cfg)
