"""Small-map layers of the tuned table: tiled kernel (its tuned configuration) vs the K-split small-map kernel, device time.
usage: python tools/bench_smallmap.py [max_positions=4096] [--write]   (--write: store "smallmap" in conv_tune.json where it wins by > 5 %)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H

MAXP = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
WRITE = "--write" in sys.argv
PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vspbfr_amd", "conv_tune.json")
table = json.load(open(PATH))
SM = H.CONFIG_IDS["smallmap"]


def timeit(fn, n=60):
    for _ in range(4):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


wins, rows = {}, []
for key, name in sorted(table.items()):
    parts = key.split(",")
    flags = [p for p in parts[12:]]
    if "t" in flags or "s" in flags:
        continue
    B, Cin, Hh, Ww, G, cg, kh, kw, st, d0, OH, OW = map(int, parts[:12])
    gs = next((int(f[1:]) for f in flags if f.startswith("g")), 0)
    if B * OH * OW > MAXP or Cin % 16 or name == "smallmap":
        continue
    if G == 4 and gs == 0:
        dil = (1, 2, 4, 8)
    elif G > 1 and gs == 0:
        continue
    else:
        dil = (d0,)
    pad = tuple(((OH - 1) * st + d * (kh - 1) + 1 - Hh + 1) // 2 for d in dil)
    if min(pad) < 0:
        continue
    x = torch.randn(B, Cin + (G - 1) * gs, Hh, Ww, device="cuda")
    wp = torch.randn(G, kh * kw, Cin, cg, device="cuda") * 0.02
    pc = H.PackedConv(wp, G, cg, Cin, kh, kw, st, dil, pad, x_group_stride=gs)
    if H.conv2d_out_size(Hh, Ww, pc) != (OH, OW) and len(dil) == 1:
        continue
    s_in = torch.rand(B, x.shape[1], device="cuda") + 0.5
    wino = name == "winograd"
    hint = 0 if wino else H.CONFIG_IDS.get(name, 0)
    try:
        t_old = timeit(lambda: H.conv2d_packed(x, pc, in_scale=s_in, tile_hint=hint, winograd=wino, bf16=False))
        t_new = timeit(lambda: H.conv2d_packed(x, pc, in_scale=s_in, tile_hint=SM, winograd=False, bf16=False))
        ya = H.conv2d_packed(x, pc, in_scale=s_in, tile_hint=hint, winograd=wino, bf16=False)
        yb = H.conv2d_packed(x, pc, in_scale=s_in, tile_hint=SM, winograd=False, bf16=False)
    except RuntimeError as ex:
        print(key, "skipped:", str(ex)[:80])
        continue
    err = (ya - yb).abs().max().item() / (ya.abs().max().item() + 1e-20)
    ok = err < 1e-4
    rows.append((t_old - t_new, key))
    print(f"{key:48s} {name:28s} {t_old:8.1f} us   smallmap {t_new:8.1f} us   {'WIN' if t_new < 0.95 * t_old else ''}  rel diff {err:.1e}{'' if ok else '  MISMATCH'}")
    if ok and t_new < 0.95 * t_old:
        wins[key] = "smallmap"
print(f"{len(wins)} wins; summed gain {sum(max(r[0], 0) for r in rows) / 1e3:.2f} ms over the listed launches (one each)")
if WRITE and wins:
    table.update(wins)
    json.dump(table, open(PATH, "w"), indent=0, sort_keys=True)
    print("written", PATH)
