"""Round 5: which Winograd entries of vspbfr_amd/conv_tune.json the fused F(4x4) kernel (vsp_conv2d_winograd4f_f32) takes over.
For every G = 1, dilation-1, stride-1 3x3 key without an affine input (no ',s' / ',t' suffix): time the table's current choice against
winograd=5 with the StyledConv operand set (style scale, demodulation, noise, bias + activation); --write rewrites the entries the fused
kernel wins by more than 3 %.  usage: tools/tune_wino4f.py [--write] [--all]   (--all: also keys the table gives to a direct kernel)"""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vspbfr_amd", "conv_tune.json")
table = json.load(open(path))
def t(f, n=5):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1000
changed = {}
for key in sorted(table, key=lambda k: [int(v) for v in k.split(",")[:12]]):
    f = key.split(",")
    if len(f) != 12: continue
    B, Cin, Hh, Ww, G, cg, kh, kw, st, dil, OH, OW = (int(v) for v in f)
    cur = table[key]
    if G != 1 or kh != 3 or kw != 3 or st != 1 or dil != 1: continue
    if "--all" not in sys.argv and not cur.startswith("winograd"): continue
    if B * Cin * Hh * Ww * 4 > 6e9: continue
    torch.manual_seed(0)
    x = torch.randn(B, Cin, Hh, Ww, device="cuda")
    w = torch.randn(cg, Cin, 3, 3, device="cuda") / math.sqrt(Cin * 9)
    pc = H.PackedConv(H.pack_weight(w), 1, cg, Cin, 3, 3, 1, (1,), (1,))
    if not H.winograd4f_eligible(pc, Hh, Ww, OH, OW): continue
    kwargs = dict(in_scale=torch.rand(B, Cin, device="cuda") + 0.5, out_scale=torch.rand(B, cg, device="cuda") + 0.5,
                  noise=torch.randn(B, 1, Hh, Ww, device="cuda"), noise_w=torch.tensor([0.3], device="cuda"), bias2=torch.randn(cg, device="cuda"), act2=1)
    wn_cur = {"winograd": True, "winograd4": 4, "winograd4f": 5}.get(cur, False)
    th = 0 if wn_cur else H.CONFIG_IDS.get(cur, 0)
    us_cur = t(lambda: H.conv2d_packed(x, pc, winograd=wn_cur, tile_hint=th, **kwargs))
    us_f = t(lambda: H.conv2d_packed(x, pc, winograd=5, **kwargs))
    us_cur = min(us_cur, t(lambda: H.conv2d_packed(x, pc, winograd=wn_cur, tile_hint=th, **kwargs)))
    us_f = min(us_f, t(lambda: H.conv2d_packed(x, pc, winograd=5, **kwargs)))
    d = (H.conv2d_packed(x, pc, winograd=5, **kwargs) - H.conv2d_packed(x, pc, winograd=False, **kwargs)).abs().max().item()
    win = us_f < 0.97 * us_cur
    print(f"{key}: {cur} {us_cur:.0f} us | winograd4f {us_f:.0f} us | max diff vs direct {d:.1e} {'<- fused' if win else ''}", flush=True)
    if win and cur != "winograd4f": changed[key] = "winograd4f"
print(len(changed), "entries to the fused kernel")
if "--write" in sys.argv and changed:
    table.update(changed)
    json.dump(table, open(os.path.join("gpurun_out", "conv_tune_w4f.json"), "w"), indent=0, sort_keys=True)
    print("wrote gpurun_out/conv_tune_w4f.json")
