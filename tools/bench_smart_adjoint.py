"""Data gradient of the four dilated SMART branches: one pass (dil_by_input_quarter, conv_pipe.hip MODE 3) vs the grouped convolution +
sum over the branches, on the training shapes (B = 4).  usage: python tools/bench_smart_adjoint.py [config name for the one-pass form]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
dev = torch.device("cuda", 0)
B = int(os.environ.get("B", 4))
hint = 0
if len(sys.argv) > 1:
    names = [H.lib.vsp_conv2d_config_name(i).decode() for i in range(H.lib.vsp_conv2d_num_configs())]
    hint = names.index(sys.argv[1]) + 1
rates = (1, 2, 4, 8)


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for cin, size in [(64, 512), (128, 256), (256, 128), (512, 64), (512, 32), (512, 16)]:
    cg = cin // 4
    g = torch.randn(B, cin, size, size, device=dev)          # gradient of the 4 x cg branch outputs
    dm = torch.rand(B, cin, device=dev) + 0.5
    ws = [torch.randn(cg, cin, 3, 3, device=dev) * 0.05 for _ in rates]
    wp = H.pack_weight_stack(ws, adjoint=True, flip=True)
    grouped = H.PackedConv(wp, 4, cin, cg, 3, 3, 1, rates, rates, x_group_stride=cg)
    a = wp.permute(1, 0, 2, 3).reshape(9, 4 * cg, cin)
    w2 = a.view(9, 4 * cg, 4, cin // 4).permute(2, 0, 1, 3).contiguous()
    one = H.PackedConv(w2, 4, cin // 4, 4 * cg, 3, 3, 1, rates, rates, dil_by_input_quarter=True)
    f_old = lambda: H.conv2d_packed(g, grouped, in_scale=dm).view(B, 4, cin, size, size).sum(1)
    f_new = lambda: H.conv2d_packed(g, one, in_scale=dm, tile_hint=hint)
    d = (f_old() - f_new()).abs().max().item()
    t_old, t_new = timeit(f_old), timeit(f_new)
    fl = 2.0 * B * cin * size * size * cin * 9
    print(f"{cin} ch @{size}: grouped + sum {t_old*1e3:.0f} us | one pass {t_new*1e3:.0f} us ({fl/t_new/1e9:.1f} TF)  max diff {d:.2e}")
