"""Autotune the conv tile configuration for the layer shapes of ONE TRAINING ITERATION (forward, data-gradient and loss-network
launches at the training batch size; GPU box).  Micro-benchmark stage of tools/autotune_conv.py over the recorded shapes; winners
that beat the cost model's pick by > 3 % are MERGED into gpurun_out/conv_tune_train.json (copy its entries into
vspbfr_amd/conv_tune.json to ship them).  usage: python tools/autotune_train.py [B] [losses] [stage_b [size]]"""
import copy, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench
from autotune_conv import timeit, WINO_ID
from vspbfr_amd import hip_ops as H
from vspbfr_amd._lib import lib
from vspbfr_amd.discriminator import Discriminator
from vspbfr_amd.train_step import RestorationTrainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
pipe = bench.build_pipeline(dev, 4, False)
G = pipe.generator
torch.manual_seed(1)
D = Discriminator(512).to(dev)
kw = {}
if len(sys.argv) > 2 and sys.argv[2] == "losses":
    from vspbfr_amd.id_loss import IDLoss
    from vspbfr_amd.lpips import PerceptualLoss
    kw = dict(percept_loss=PerceptualLoss().to(dev), percept_weight=0.5, id_loss=IDLoss(None, device=dev), id_weight=0.1)
STAGE_B = len(sys.argv) > 3 and sys.argv[3] == "stage_b"      # code_diffuser_train.py iteration (size argv[4], default 256)
if STAGE_B:
    from vspbfr_amd.train_step import CodeDiffuserTrainer
    size = int(sys.argv[4]) if len(sys.argv) > 4 else 256
    pipe.psp.E4Enet.out_size = size
    trb = CodeDiffuserTrainer(pipe.diffusion, pipe.psp, percept_loss=kw.get("percept_loss"), id_loss=kw.get("id_loss"))
    low, real = torch.rand(B, 3, size, size, device=dev) * 2 - 1, torch.rand(B, 3, size, size, device=dev) * 2 - 1
    trb.step(low, real)
    H.RECORDER = []
    trb.step(low, real)
else:
    tr = RestorationTrainer(G, copy.deepcopy(G), D, psp_embedding=pipe.psp, diffusion=pipe.diffusion, mixing=0.9, **kw)
    low, real = torch.rand(B, 3, 512, 512, device=dev) * 2 - 1, torch.rand(B, 3, 512, 512, device=dev) * 2 - 1
    G.train()
    tr.step(1, low, real)
    H.RECORDER = []
    tr.step(16, low, real)          # an iteration with the R1 pass
torch.cuda.synchronize()
recs, H.RECORDER = H.RECORDER, None
uniq = {}
for key, dims, pc, trp in recs:
    uniq.setdefault(key, [dims, pc, 0, trp])[2] += 1
shipped = H._load_tune_table()
print(f"{len(recs)} conv launches, {len(uniq)} distinct shapes, {sum(k in shipped for k in uniq)} already tuned", flush=True)
n = lib.vsp_conv2d_num_configs()
table, report = {}, []
tot0 = tot1 = 0.0
for key, (dims, pc, count, trp) in uniq.items():
    if key in shipped and not os.environ.get("RETUNE"):   # RETUNE=1: measure the shipped entries again (new kernel families)
        continue
    Bq, Cin, Hh, Ww, OH, OW = dims
    x = torch.randn(Bq, (pc.G - 1) * pc.x_group_stride + Cin, Hh, Ww, device=dev)
    shift = torch.zeros(Cin, device=dev)
    out = torch.empty(Bq, pc.cout, max(OH, 1) * 2 + 1, max(OW, 1) * 2 + 1, device=dev)
    times = {}
    ref = None      # every candidate must reproduce the cost model's pick (a configuration that runs is not yet one that is right
    bad = []        # for a shape nobody tested it on: ragged channel counts, 4x4 maps, 4x4 kernels)
    for c in range(0, n + 1):
        try:
            kwc = dict(transposed=True) if trp else dict(out=out, n_out=(OH, OW))
            if key.endswith(",s"):
                kwc["in_shift"] = shift
            out.zero_()
            y = H.conv2d_packed(x, pc, tile_hint=c, winograd=False, **kwc)
            y = (y if trp else y[:, :, :OH, :OW]).clone()
            if c == 0:
                ref = y
            elif ref is not None and float((y - ref).abs().max()) > 1e-4 * float(ref.abs().max()) + 1e-6:
                bad.append(c)
                continue
            est = timeit(lambda: H.conv2d_packed(x, pc, tile_hint=c, winograd=False, **kwc), 1)
            times[c] = timeit(lambda: H.conv2d_packed(x, pc, tile_hint=c, winograd=False, **kwc), 3 if est > 0.3 else 10)
        except RuntimeError:
            continue
    if not trp and H.winograd_eligible(pc, Hh, Ww, OH, OW):
        try:
            kwc = dict(out=out, n_out=(OH, OW), winograd=True)
            out.zero_()
            y = H.conv2d_packed(x, pc, **kwc)[:, :, :OH, :OW]
            if ref is not None and float((y - ref).abs().max()) > 1e-4 * float(ref.abs().max()) + 1e-6:
                bad.append("winograd")
            else:
                times[WINO_ID] = timeit(lambda: H.conv2d_packed(x, pc, **kwc), 5)
        except RuntimeError:
            pass
    if bad:
        print("WRONG RESULT:", key, [c if isinstance(c, str) else lib.vsp_conv2d_config_name(c - 1).decode() for c in bad], flush=True)
    if 0 not in times or len(times) < 2:
        continue
    best = min((c for c in times if c > 0), key=lambda c: times[c])
    # what the library does today for this shape: the cost model (or the Winograd decision inherited from the batch-8 entry)
    t_now = times[0]
    tot0 += t_now * count
    name = "winograd" if best == WINO_ID else lib.vsp_conv2d_config_name(best - 1).decode()
    if times[best] < 0.97 * t_now:
        table[key] = name
        tot1 += times[best] * count
    else:
        tot1 += t_now * count
    report.append(((t_now - min(times[best], t_now)) * count, key, count, name, round(times[best] * 1e3, 1), round(t_now * 1e3, 1)))
report.sort(reverse=True)
for r in report[:40]:
    print("saves %.2f ms | %s | x%d | best %s %s us (cost-model pick %s us)" % r, flush=True)
print(f"conv time of the recorded launches: cost model {tot0:.1f} ms -> tuned {tot1:.1f} ms; {len(table)} new entries")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(table, open("gpurun_out/conv_tune_train.json", "w"), indent=0, sort_keys=True)
