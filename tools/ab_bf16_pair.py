"""bf16-activation conv kernel (vsp_conv2d_bf16, io_bf16): pixel-pair staging against the one-pixel tasks (env VSP_BF16_PAIR=0) -- bit-identical
outputs over every mode / tile variant, and the time of both.  usage: python tools/ab_bf16_pair.py   (runs itself twice as child processes)"""
import math, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = [  # (mode, B, Cin, Cout, H, W, G, dils, hint)
    ("s1", 4, 64, 64, 128, 128, 1, (1,), 0), ("s1", 2, 512, 512, 64, 64, 1, (1,), 0), ("s1", 2, 128, 128, 96, 160, 1, (1,), 4), ("s1", 2, 256, 96, 32, 32, 1, (1,), 1),
    ("s1", 2, 128, 32, 64, 64, 4, (1, 2, 4, 8), 0), ("s1", 2, 256, 64, 32, 48, 4, (1, 2, 4, 8), 0), ("s1", 3, 64, 40, 20, 36, 1, (1,), 7), ("s1", 2, 32, 32, 16, 8, 1, (1,), 0),
    ("s2", 2, 64, 128, 64, 64, 1, (1,), 0), ("s2", 2, 512, 1024, 32, 32, 1, (1,), 3), ("s2", 2, 128, 256, 66, 66, 1, (1,), 6), ("s2p0", 2, 64, 128, 66, 66, 1, (1,), 0),
    ("tc", 2, 64, 32, 64, 64, 1, (1,), 0), ("tc", 2, 128, 64, 32, 48, 1, (1,), 4), ("tc", 2, 512, 512, 16, 16, 1, (1,), 8),
]
BIG = [("s1", 16, 512, 512, 64, 64, 1, (1,), 0), ("s1", 16, 128, 32, 256, 256, 4, (1, 2, 4, 8), 0), ("s2", 16, 512, 512, 64, 64, 1, (1,), 0), ("tc", 16, 128, 64, 256, 256, 1, (1,), 0),
       ("tc", 16, 512, 256, 64, 64, 1, (1,), 0), ("s1", 16, 256, 256, 32, 32, 1, (1,), 0)]
if os.environ.get("QUICK"):
    BIG = []
def child(tag):
    import torch
    from vspbfr_amd import hip_ops as H
    def t(f, n=10):
        f(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): f()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1000
    outs, times = [], []
    for k, (mode, B, Cin, Cout, Hh, Ww, G, dils, hint) in enumerate(CASES + BIG):
        g_ = torch.Generator().manual_seed(k)
        x = torch.randn(B, Cin, Hh, Ww, generator=g_).cuda().to(torch.bfloat16)
        sc = (torch.rand(B, Cin, generator=g_) + 0.5).cuda()
        if G > 1:
            wp = (torch.randn(G, 9, Cin, Cout, generator=g_) / math.sqrt(Cin * 9)).cuda()
            pc = H.PackedConv(wp, G, Cout, Cin, 3, 3, 1, dils, dils)
            f = lambda: H.conv2d_packed(x, pc, in_scale=sc, bf16=True, tile_hint=hint)
        else:
            w = (torch.randn(Cout, Cin, 3, 3, generator=g_) / math.sqrt(Cin * 9)).cuda()
            if mode == "tc":
                pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, 3, 3, 1, (1,), (1,))
                f = lambda: H.conv_transpose2d_s2_fused(x, pc, in_scale=sc, bf16=True, tile_hint=hint)
            else:
                stride, pad = (2, 1) if mode == "s2" else ((2, 0) if mode == "s2p0" else (1, 1))
                pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, 3, 3, stride, (1,), (pad,))
                f = lambda: H.conv2d_packed(x, pc, in_scale=sc, bf16=True, tile_hint=hint)
        y = f()
        assert y.dtype == torch.bfloat16
        outs.append(y.float().cpu())
        times.append(t(f) if k >= len(CASES) else 0.0)
    torch.save({"outs": outs, "times": times}, f"/tmp/ab_bf16_pair_{tag}.pt")
if len(sys.argv) > 1:
    child(sys.argv[1])
else:
    import torch
    # AB_LIBS=<a.so>,<b.so>: the same comparison between two builds of the library instead of the two staging forms
    libs = os.environ.get("AB_LIBS", "").split(",") if os.environ.get("AB_LIBS") else None
    for k, (tag, env) in enumerate((("pair", "1"), ("one", "0"))):
        e = dict(os.environ, VSP_TUNE="1", VSP_BF16_MODW="0", VSP_BF16_PAIR=env) if libs is None else dict(os.environ, VSPBFR_HIP_LIB=libs[k])
        subprocess.check_call([sys.executable, os.path.abspath(__file__), tag], env=e)
    a, b = torch.load("/tmp/ab_bf16_pair_pair.pt"), torch.load("/tmp/ab_bf16_pair_one.pt")
    bad = 0
    for k, (ya, yb) in enumerate(zip(a["outs"], b["outs"])):
        same = torch.equal(ya, yb)
        bad += not same
        case = (CASES + BIG)[k]
        extra = f" | {'first library' if libs else 'pair'} {a['times'][k]:.0f} us, {'second library' if libs else 'one-pixel tasks'} {b['times'][k]:.0f} us" if k >= len(CASES) else ""
        print(f"{case}: {'bit-identical' if same else 'DIFFERENT max ' + str((ya - yb).abs().max().item())}{extra}", flush=True)
    sys.exit(1 if bad else 0)
