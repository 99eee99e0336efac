"""Single dilated 3x3 conv (one SMART branch) per dilation: where does the grouped launch lose time?"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
names = sys.argv[1:] or ["0"]
for (B, Cin, Cg, S) in [(8, 256, 64, 128), (8, 128, 32, 256)]:
    x = torch.randn(B, Cin, S, S, device="cuda")
    for d in (1, 2, 4, 8):
        w = torch.randn(Cg, Cin, 3, 3, device="cuda") / math.sqrt(Cin * 9)
        pc = H.PackedConv(H.pack_weight(w), 1, Cg, Cin, 3, 3, 1, (d,), (d,))
        out = []
        for nm in names:
            cfg = 0 if nm == "0" else H.CONFIG_IDS[nm]
            f = lambda: H.conv2d_packed(x, pc, tile_hint=cfg)
            try: f()
            except RuntimeError: out.append(f"{nm}: n/a"); continue
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): f()
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) * 200
            out.append(f"{nm}: {us:.0f} us {2.0*B*Cg*Cin*9*S*S/us/1e6:.1f} TF")
        print(f"{Cin}->{Cg} @{S} d={d}: " + " | ".join(out))
