import math, sys
sys.path.insert(0, "/root/repo")
import torch
from vspbfr_amd import hip_ops as H
def t(f, n=10):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1000
for (B, Cin, S) in ((16, 64, 256), (4, 64, 512)):
    x = torch.randn(B, Cin, S, S, device="cuda")
    w = torch.randn(3, Cin, 3, 3, device="cuda") / math.sqrt(Cin * 9)
    pc = H.PackedConv(H.pack_weight(w), 1, 3, Cin, 3, 3, 1, (1,), (1,))
    b = torch.randn(3, device="cuda")
    r = {}
    for name, kw in (("winograd4f", dict(winograd=5)), ("F(2x2)", dict(winograd=True)), ("direct / cost model", dict(winograd=False))):
        r[name] = t(lambda: H.conv2d_packed(x, pc, ch_bias=b, **kw))
    print(f"{B} x {Cin} -> 3 at {S}^2:", {k: round(v, 1) for k, v in r.items()})
