"""The shared-input dilation-group launches of the bf16 kernel, fp32 and bf16 I/O (the region-major work order: VSP_CONV_DBG=8388608
with the ablation library switches it off)."""
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
B = int(os.environ.get("B", 16))
for (Cin, Cg, S) in [(64, 16, 512), (128, 32, 256), (256, 64, 128), (512, 128, 64), (512, 128, 32)]:
    x = torch.randn(B, Cin, S, S, device="cuda")
    wp = torch.randn(4, 9, Cin, Cg, device="cuda") / math.sqrt(Cin * 9)
    pc = H.PackedConv(wp, 4, Cg, Cin, 3, 3, 1, (1, 2, 4, 8), (1, 2, 4, 8))
    sc = torch.rand(B, Cin, device="cuda") + 0.5
    fl = 2.0 * B * 4 * Cg * Cin * 9 * S * S
    xb = x.to(torch.bfloat16)
    u32 = t(lambda: H.conv2d_packed(x, pc, in_scale=sc, bf16=True))
    u16 = t(lambda: H.conv2d_packed(xb, pc, in_scale=sc, bf16=True))
    print(f"dbg={os.environ.get('VSP_CONV_DBG', '0'):>8} {Cin}->4x{Cg} @{S} B={B}: fp32 I/O {u32:.0f} us {fl/u32/1e6:.0f} TF | bf16 I/O {u16:.0f} us {fl/u16/1e6:.0f} TF", flush=True)
