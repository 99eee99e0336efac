"""Time the four-dilation SMART branch conv on the path's shapes.  usage: bench_dgconv.py [cfg-name ...]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
names = sys.argv[1:] or ["0"]
DILS = tuple(int(v) for v in os.environ.get("DILS", "1,2,4,8").split(","))
for (B, Cin, Cg, S) in [(8, 64, 16, 512), (8, 128, 32, 256), (8, 256, 64, 128), (8, 512, 128, 64), (8, 512, 128, 32)]:
    x = torch.randn(B, Cin, S, S, device="cuda")
    wp = torch.randn(4, 9, Cin, Cg, device="cuda") / math.sqrt(Cin * 9)
    pc = H.PackedConv(wp, 4, Cg, Cin, 3, 3, 1, DILS, DILS)
    sc = torch.rand(B, Cin, device="cuda") + 0.5
    out = []
    for nm in names:
        cfg = 0 if nm == "0" else H.CONFIG_IDS[nm]
        f = lambda: H.conv2d_packed(x, pc, in_scale=sc, tile_hint=cfg)
        try:
            f()
        except RuntimeError as ex:
            out.append(f"{nm}: n/a"); continue
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): f()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 200
        out.append(f"{nm}: {us:.0f} us {2.0*B*4*Cg*Cin*9*S*S/us/1e6:.1f} TF")
    print(f"{Cin}->4x{Cg} @{S}: " + " | ".join(out))
