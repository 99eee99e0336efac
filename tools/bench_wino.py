"""Winograd F(2x2,3x3) vs the direct kernel on the stride-1 3x3 layers of the path."""
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
for (B, Cin, Cout, S) in [(8, 64, 64, 512), (8, 128, 128, 256), (8, 256, 256, 128), (8, 512, 512, 64), (8, 512, 512, 32), (8, 256, 256, 32),
                          (8, 32, 32, 1024), (8, 64, 64, 128), (8, 512, 512, 16)]:
    x = torch.randn(B, Cin, S, S, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(Cin * 9)
    pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, 3, 3, 1, (1,), (1,))
    sc = torch.rand(B, Cin, device="cuda") + 0.5
    fl = 2.0 * B * Cout * Cin * 9 * S * S
    ud = t(lambda: H.conv2d_packed(x, pc, in_scale=sc, winograd=False))
    uw = t(lambda: H.conv2d_packed(x, pc, in_scale=sc, winograd=True))
    err = (H.conv2d_packed(x, pc, in_scale=sc, winograd=True) - H.conv2d_packed(x, pc, in_scale=sc, winograd=False)).abs().max().item()
    print(f"{Cin}->{Cout} @{S}: direct {ud:.0f} us {fl/ud/1e6:.1f} TF | winograd {uw:.0f} us {fl/uw/1e6:.1f} eff. TF | x{ud/uw:.2f} | max diff {err:.2e}")

if len(sys.argv) > 1 and sys.argv[1] == "plain":
    sys.exit(0)
print("dilation groups (1, 2, 4, 8):")
for (B, Cin, Cg, S) in [(8, 64, 16, 512), (8, 128, 32, 256), (8, 256, 64, 128), (8, 512, 128, 64), (8, 512, 128, 32)]:
    x = torch.randn(B, Cin, S, S, device="cuda")
    wp = torch.randn(4, 9, Cin, Cg, device="cuda") / math.sqrt(Cin * 9)
    pc = H.PackedConv(wp, 4, Cg, Cin, 3, 3, 1, (1, 2, 4, 8), (1, 2, 4, 8))
    sc = torch.rand(B, Cin, device="cuda") + 0.5
    fl = 2.0 * B * 4 * Cg * Cin * 9 * S * S
    ud = t(lambda: H.conv2d_packed(x, pc, in_scale=sc, winograd=False))
    uw = t(lambda: H.conv2d_packed(x, pc, in_scale=sc, winograd=True))
    err = (H.conv2d_packed(x, pc, in_scale=sc, winograd=True) - H.conv2d_packed(x, pc, in_scale=sc, winograd=False)).abs().max().item()
    print(f"{Cin}->4x{Cg} @{S}: direct {ud:.0f} us {fl/ud/1e6:.1f} TF | winograd {uw:.0f} us {fl/uw/1e6:.1f} eff. TF | x{ud/uw:.2f} | max diff {err:.2e}")
