"""bf16-MFMA conv kernel (vsp_conv2d_bf16, every tile variant) vs the tuned fp32 kernels on the stride-1 3x3 layers of the path."""
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
X3 = bool(os.environ.get("X3"))   # X3=1: the split-precision form (vsp_conv2d_bf16x3, variants 1 / 4 / 7)
def run(x, pc, sc, fl, tag):
    uf = t(lambda: H.conv2d_packed(x, pc, in_scale=sc))
    ref = H.conv2d_packed(x, pc, in_scale=sc)
    out = [f"{tag}: fp32 {uf:.0f} us {fl/uf/1e6:.0f} TF |"]
    for v in ((0, 1, 4, 7) if X3 else (0, 1, 2, 3, 4, 6, 7)):
        try:
            ub = t(lambda: H.conv2d_packed(x, pc, in_scale=sc, bf16="x3" if X3 else True, tile_hint=v))
        except RuntimeError as ex:
            out.append(f" v{v} n/a"); continue
        out.append(f" v{v} {ub:.0f} us {fl/ub/1e6:.0f} TF x{uf/ub:.2f} |")
    err = (H.conv2d_packed(x, pc, in_scale=sc, bf16="x3" if X3 else True) - ref).abs().max().item() / ref.abs().max().item()
    print("".join(out) + f" rel err {err:.1e}", flush=True)
B = int(os.environ.get("B", 8))
for (Cin, Cout, S) in [(64, 64, 512), (128, 128, 256), (256, 256, 128), (512, 512, 64), (512, 512, 32), (256, 256, 32),
                       (32, 32, 1024), (64, 64, 128), (128, 128, 64), (512, 512, 16)]:
    x = torch.randn(B, Cin, S, S, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(Cin * 9)
    pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, 3, 3, 1, (1,), (1,))
    sc = torch.rand(B, Cin, device="cuda") + 0.5
    run(x, pc, sc, 2.0 * B * Cout * Cin * 9 * S * S, f"{Cin}->{Cout} @{S}")
print("dilation groups (1, 2, 4, 8):")
for (Cin, Cg, S) in [(64, 16, 512), (128, 32, 256), (256, 64, 128), (512, 128, 64), (512, 128, 32)]:
    x = torch.randn(B, Cin, S, S, device="cuda")
    wp = torch.randn(4, 9, Cin, Cg, device="cuda") / math.sqrt(Cin * 9)
    pc = H.PackedConv(wp, 4, Cg, Cin, 3, 3, 1, (1, 2, 4, 8), (1, 2, 4, 8))
    sc = torch.rand(B, Cin, device="cuda") + 0.5
    run(x, pc, sc, 2.0 * B * 4 * Cg * Cin * 9 * S * S, f"{Cin}->4x{Cg} @{S}")
