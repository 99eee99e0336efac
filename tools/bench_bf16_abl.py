"""Ablations of the bf16 conv kernel (VSP_CONV_DBG: 1 = stage the first chunk only, 2 = no MFMA phase, 4 = no epilogue)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
def t(f, n=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1000
B = 8
for (Cin, Cout, S, v) in [(64, 64, 512, 2), (64, 64, 512, 4), (128, 128, 256, 2), (256, 256, 128, 2), (512, 512, 64, 2), (512, 512, 64, 3), (512, 512, 32, 1)]:
    x = torch.randn(B, Cin, S, S, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(Cin * 9)
    pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, 3, 3, 1, (1,), (1,))
    sc = torch.rand(B, Cin, device="cuda") + 0.5
    fl = 2.0 * B * Cout * Cin * 9 * S * S
    u = t(lambda: H.conv2d_packed(x, pc, in_scale=sc, bf16=True, tile_hint=v))
    print(f"dbg={os.environ.get('VSP_CONV_DBG','0')} {Cin}->{Cout} @{S} v{v}: {u:.0f} us {fl/u/1e6:.0f} TF", flush=True)
