"""Round 3: the pipelined direct kernels (conv_pipe.hip, configuration names "...p3...") against the tuned table's choice on the
layers that have no Winograd form.  usage: bench_pipe.py [shape,shape...]   (GPU box)"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
from vspbfr_amd._lib import lib

SHAPES = {  # name: B, Cin, Cout, H, W, stride, pad, G (4 = dilation groups 1,2,4,8), transposed
    "stem": (8, 512, 5632, 64, 64, 2, 1, 1, False),
    "down256": (8, 64, 128, 513, 513, 2, 0, 1, False),
    "down128": (8, 128, 256, 257, 257, 2, 0, 1, False),
    "down64": (8, 256, 512, 129, 129, 2, 0, 1, False),
    "down32": (8, 512, 512, 65, 65, 2, 0, 1, False),
    "head2048": (8, 512, 2048, 32, 32, 2, 1, 1, False),
    "dil512": (8, 64, 64, 512, 512, 1, 0, 4, False),
    "up64": (8, 512, 256, 64, 64, 1, 0, 1, True),
    "up128": (8, 256, 128, 128, 128, 1, 0, 1, True),
    "up256": (8, 128, 64, 256, 256, 1, 0, 1, True),
    "up512": (8, 64, 32, 512, 512, 1, 0, 1, True),
    "up32": (8, 512, 512, 32, 32, 1, 0, 1, True),
    "s1_512": (8, 512, 512, 64, 64, 1, 1, 1, False),
    "narrow1024": (8, 32, 32, 1024, 1024, 1, 1, 1, False),
}


def t(f, n=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1000


def main():
    only = sys.argv[1].split(",") if len(sys.argv) > 1 else list(SHAPES)
    names = [lib.vsp_conv2d_config_name(i).decode() for i in range(lib.vsp_conv2d_num_configs())]
    pipe = [i + 1 for i, n in enumerate(names) if "p3" in n]
    for name in only:
        B, Cin, Cout, Hh, Ww, s, pad, G, tr = SHAPES[name]
        x = torch.randn(B, Cin, Hh, Ww, device="cuda")
        sc = torch.rand(B, Cin, device="cuda") + 0.5
        if G == 1:
            w = torch.randn(Cout, Cin, 3, 3, device="cuda") / math.sqrt(Cin * 9)
            pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, 3, 3, s, (1,), (pad,))
        else:
            wp = torch.randn(4, 9, Cin, Cout // 4, device="cuda") / math.sqrt(Cin * 9)
            pc = H.PackedConv(wp, 4, Cout // 4, Cin, 3, 3, 1, (1, 2, 4, 8), (1, 2, 4, 8))
        kw = dict(in_scale=sc, transposed=tr, winograd=False, bf16=False)
        ref = H.conv2d_packed(x, pc, **kw)
        oh, ow = (Hh, Ww) if tr else H.conv2d_out_size(Hh, Ww, pc)
        fl = 2.0 * B * Cout * oh * ow * Cin * 9
        us = t(lambda: H.conv2d_packed(x, pc, **kw))
        print(f"{name}: tuned {us:.0f} us {fl/us/1e6:.1f} TF", flush=True)
        for c in pipe:
            try:
                y = H.conv2d_packed(x, pc, tile_hint=c, **kw)
            except RuntimeError as ex:
                continue
            err = (y - ref).abs().max().item()
            us = t(lambda: H.conv2d_packed(x, pc, tile_hint=c, **kw))
            print(f"    {names[c-1]}: {us:.0f} us {fl/us/1e6:.1f} TF  max diff {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
