"""bf16 kernel on the stride-2 and transposed layers of the path: every tile variant vs the tuned fp32 kernel."""
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
print("stride 2:")
for (Cin, Cout, S, pad, G) in [(512, 5632, 64, 1, 1), (64, 128, 513, 0, 1), (128, 256, 257, 0, 1), (256, 512, 129, 0, 1), (512, 512, 65, 0, 1), (512, 512, 32, 1, 11), (512, 2048, 32, 1, 1)]:
    x = torch.randn(B, Cin * (G if G > 1 else 1), S, S, device="cuda")
    cg = Cout // G if G > 1 else Cout
    if G > 1:
        wp = torch.randn(G, 9, Cin, cg, device="cuda") / math.sqrt(Cin * 9)
        pc = H.PackedConv(wp, G, cg, Cin, 3, 3, 2, (1,), (pad,), x_group_stride=Cin)
    else:
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(Cin * 9)
        pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, 3, 3, 2, (1,), (pad,))
    oh, ow = H.conv2d_out_size(S, S, pc)
    fl = 2.0 * B * (cg * G if G > 1 else Cout) * Cin * 9 * oh * ow
    uf = t(lambda: H.conv2d_packed(x, pc))
    out = [f"{Cin}->{Cout} G{G} @{S}->{oh}: fp32 {uf:.0f} us {fl/uf/1e6:.0f} TF |"]
    for v in ((0, 6, 7) if os.environ.get("X3") else (0, 4, 6, 2, 3)):
        try:
            ub = t(lambda: H.conv2d_packed(x, pc, bf16="x3" if os.environ.get("X3") else True, tile_hint=v))
        except Exception as e:
            out.append(f" v{v} n/a |"); continue
        out.append(f" v{v} {ub:.0f} us {fl/ub/1e6:.0f} TF x{uf/ub:.2f} |")
    print("".join(out), flush=True)
print("transposed:")
for (Cin, Cout, S) in [(64, 32, 512), (128, 64, 256), (256, 128, 128), (512, 256, 64), (512, 512, 32), (512, 512, 16)]:
    x = torch.randn(B, Cin, S, S, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(Cin * 9)
    pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, 3, 3, 1, (1,), (1,))
    sc = torch.rand(B, Cin, device="cuda") + 0.5
    fl = 2.0 * B * Cout * Cin * 9 * S * S
    uf = t(lambda: H.conv_transpose2d_s2_fused(x, pc, in_scale=sc))
    out = [f"{Cin}->{Cout} @{S}->{2*S+1}: fp32 {uf:.0f} us {fl/uf/1e6:.0f} TF |"]
    for v in (0, 4, 5, 8):
        ub = t(lambda: H.conv_transpose2d_s2_fused(x, pc, in_scale=sc, bf16=True, tile_hint=v))
        out.append(f" v{v} {ub:.0f} us {fl/ub/1e6:.0f} TF x{uf/ub:.2f} |")
    print("".join(out), flush=True)
