"""Where do the memcpy / memset launches and the ATen copy_ / add kernels of one training iteration come from?  Kineto trace of one
iteration (forward thread + autograd thread), each runtime memcpy / memset call and each aten::copy_ / aten::add / aten::fill_ op
attributed to the enclosing ATen op chain and (forward thread) the innermost Python frame of this repository.
usage: python tools/find_copies_train.py [B]"""
import collections, copy, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
from vspbfr_amd.discriminator import Discriminator
from vspbfr_amd.train_step import RestorationTrainer
from vspbfr_amd.id_loss import IDLoss
from vspbfr_amd.lpips import PerceptualLoss

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
pipe = bench.build_pipeline(dev, 4, False)
G = pipe.generator
torch.manual_seed(1)
D = Discriminator(512).to(dev)
kw = dict(percept_loss=PerceptualLoss().to(dev), percept_weight=0.5, id_loss=IDLoss(None, device=dev), id_weight=0.1)
tr = RestorationTrainer(G, copy.deepcopy(G), D, psp_embedding=pipe.psp, diffusion=pipe.diffusion, mixing=0.9, **kw)
low, real = torch.rand(B, 3, 512, 512, device=dev) * 2 - 1, torch.rand(B, 3, 512, 512, device=dev) * 2 - 1
G.train()
for i in (1, 2):
    tr.step(i, low, real)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(3, low, real)
    torch.cuda.synchronize()
os.makedirs("gpurun_out", exist_ok=True)
path = "gpurun_out/find_copies_train_trace.json"
prof.export_chrome_trace(path)
trj = json.load(open(path))
evs = [e for e in trj["traceEvents"] if e.get("ph") == "X"]
rt = [e for e in evs if e.get("cat") in ("cuda_runtime", "cuda_driver") and ("emcpy" in e["name"] or "emset" in e["name"])]
ops = [e for e in evs if e.get("cat") == "cpu_op"]
py = sorted((e for e in evs if e.get("cat") == "python_function" and "vspbfr_amd" in e["name"]), key=lambda e: e["ts"])
byt = collections.defaultdict(list)
for e in ops:
    byt[e.get("tid")].append(e)


def chain(ev):
    """names of the ATen ops / autograd nodes on the same thread that enclose ev, outermost first"""
    t, tid = ev["ts"], ev.get("tid")
    enc = [e for e in byt[tid] if e["ts"] <= t and t + ev.get("dur", 0) <= e["ts"] + e["dur"] and e is not ev]
    enc.sort(key=lambda e: -e["dur"])
    return " > ".join(e["name"] for e in enc[:4])


def frame(ev):
    t = ev["ts"]
    best = None
    for e in py:
        if e["ts"] <= t <= e["ts"] + e["dur"] and e.get("tid") == ev.get("tid") and (best is None or e["dur"] < best["dur"]):
            best = e
    return best["name"].split("vspbfr_amd/")[-1] if best else "?"


cnt = collections.Counter()
for r in rt:
    cnt[(r["name"], chain(r), frame(r))] += 1
print(f"{len(rt)} runtime memcpy / memset calls")
for k, n in sorted(cnt.items(), key=lambda kv: -kv[1])[:30]:
    print(f"{n:5d}  {k}")
cnt = collections.Counter()
for o in ops:
    if o["name"] in ("aten::copy_", "aten::add", "aten::add_", "aten::fill_", "aten::zero_", "aten::clone", "aten::contiguous"):
        cnt[(o["name"], chain(o), frame(o))] += 1
print("-- ATen copy / add / fill ops by enclosing chain")
for k, n in sorted(cnt.items(), key=lambda kv: -kv[1])[:50]:
    print(f"{n:5d}  {k}")
os.remove(path)
