import sys, torch
sys.path.insert(0, '.')
from vspbfr_amd import hip_ops as H
SM = H.CONFIG_IDS["smallmap"]
def timeit(fn, n=100):
    for _ in range(4): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (B, cin, cout, hw, k, st) in [(4, 512, 512, 4, 3, 1), (8, 512, 512, 4, 3, 1), (8, 512, 512, 8, 3, 1), (4, 512, 512, 8, 3, 1), (4, 512, 512, 17, 3, 2), (8, 2048, 512, 4, 1, 1), (8, 512, 5632, 2, 3, 2)]:
    x = torch.randn(B, cin, hw, hw, device="cuda"); wp = torch.randn(1, k * k, cin, cout, device="cuda") * 0.02
    pc = H.PackedConv(wp, 1, cout, cin, k, k, st, (1,), (k // 2,))
    s_in = torch.rand(B, cin, device="cuda") + 0.5
    print((B, cin, cout, hw, k, st), "%.1f us" % timeit(lambda: H.conv2d_packed(x, pc, in_scale=s_in, tile_hint=SM, winograd=False, bf16=False)))
