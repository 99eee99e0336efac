"""Stress of vsp_conv2d_bf16rv for the packed-fp32 miscompare: N launches of one layer, count of launches / elements that
differ from the first launch and from the no-SLP production library's result.   usage: rv_stress.py [N] ; env VSPBFR_HIP_LIB selects the build"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for (B, Cin, Cout, S, hint) in ((5, 64, 64, 256, 0), (6, 32, 32, 256, 1)):
    g_ = torch.Generator().manual_seed(B * 1000 + Cin)
    x = torch.randn(B, Cin, S, S, generator=g_).to(torch.bfloat16).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g_) / math.sqrt(Cin * 9)).cuda()
    s_in, demod, bias = (torch.rand(B, Cin, generator=g_) + 0.5).cuda(), (torch.rand(B, Cout, generator=g_) + 0.5).cuda(), torch.randn(Cout, generator=g_).cuda()
    nz, nw = torch.randn(B, 1, S, S, generator=g_).cuda(), torch.full((1,), 0.2).cuda()
    pc = H.PackedConv(H.pack_weight(w), 1, Cout, Cin, 3, 3, 1, (1,), (1,))
    for name, kw in (("modulated", dict(in_scale=s_in, out_scale=demod, act2=1, bias2=bias, noise=nz, noise_w=nw)), ("bare", dict()),
                     ("in_scale only", dict(in_scale=s_in)), ("epilogue only", dict(out_scale=demod, act2=1, bias2=bias)),
                     ("out_scale only", dict(out_scale=demod)), ("noise only", dict(noise=nz, noise_w=nw))):
        first = H.conv2d_packed(x, pc, bf16="rv", tile_hint=hint, **kw)
        ref_path = f"/tmp/rv_ref_{B}_{Cin}_{name.replace(' ', '_')}.pt"
        if not os.environ.get("VSPBFR_HIP_LIB"):
            torch.save(first.cpu(), ref_path)
        elif os.path.exists(ref_path):
            ref = torch.load(ref_path).cuda()
            print(f"   first launch vs the production library: {int((first != ref).sum())} elements differ, max |d| {float((first.float() - ref.float()).abs().max()):.4g}")
        bad_launches, bad_elems, lanes = 0, 0, {}
        for _ in range(N):
            y = H.conv2d_packed(x, pc, bf16="rv", tile_hint=hint, **kw)
            d = (y != first)
            n = int(d.sum())
            if n:
                bad_launches += 1
                bad_elems += n
                xs = d.nonzero()[:, 3] % 64     # column inside the 64-pixel tile row: a lane owns the pixel pair (2 l32, 2 l32 + 1)
                for v in xs.tolist()[:2000]:
                    lanes[v // 2] = lanes.get(v // 2, 0) + 1
        print(f"{os.environ.get('VSPBFR_HIP_LIB', 'production')}: {B}x{Cin}->{Cout}@{S} {name}: {bad_launches}/{N} launches differ from the first, "
              f"{bad_elems} elements; pixel pairs (l32) hit: {dict(sorted(lanes.items()))}", flush=True)
