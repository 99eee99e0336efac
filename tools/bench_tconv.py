"""Time the one-pass transposed conv on the up-conv shapes of the path.  usage: bench_tconv.py [cfg-name]  (env VSP_CONV_DBG for ablations)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
cfg = H.CONFIG_IDS[sys.argv[1]] if len(sys.argv) > 1 else 0
for (B, Cin, Cout, S) in [(8, 512, 256, 64), (8, 256, 128, 128), (8, 128, 64, 256), (8, 64, 32, 512), (8, 512, 512, 32)]:
    x = torch.randn(B, Cin, S, S, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(Cin * 9)
    pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, 3, 3, 1, (1,), (1,))
    sc = torch.rand(B, Cin, device="cuda") + 0.5
    f = lambda: H.conv_transpose2d_s2_fused(x, pc, in_scale=sc, out_scale=None, tile_hint=cfg)
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 100
    print(f"{Cin}->{Cout} @{S}: {us:.0f} us  {2.0*B*Cout*Cin*9*S*S/us/1e6:.1f} TF")
