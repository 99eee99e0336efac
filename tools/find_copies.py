"""Round 3: where do the __amd_rocclr_copyBuffer launches of an inference step come from?  Profiles one pipeline pass with the
kineto tracer (CPU + device activities), then attributes every memcpy runtime call to the innermost Python frame of this
repository that encloses it.  usage: find_copies.py   (GPU box; env B, T)"""
import collections, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
pipe = bench.build_pipeline(dev, int(os.environ.get("T", 50)), True, noise_seed=123)
lq = torch.rand(int(os.environ.get("B", 8)), 3, 512, 512, device=dev) * 2 - 1
with torch.no_grad():
    pipe(lq)
    pipe(lq)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    with torch.no_grad():
        pipe(lq)
    torch.cuda.synchronize()
os.makedirs("gpurun_out", exist_ok=True)
path = "gpurun_out/find_copies_trace.json"
prof.export_chrome_trace(path)
tr = json.load(open(path))
evs = [e for e in tr["traceEvents"] if e.get("ph") == "X"]
names = collections.Counter(e["name"] for e in evs if "emcpy" in e["name"] or "emset" in e["name"] or "copyBuffer" in e["name"])
print("memcpy-like events:", dict(names))
rt = [e for e in evs if e.get("cat") in ("cuda_runtime", "cuda_driver") and ("emcpy" in e["name"] or "emset" in e["name"])]
py = sorted((e for e in evs if e.get("cat") == "python_function" and ("vspbfr_amd" in e["name"] or "bench.py" in e["name"])), key=lambda e: e["ts"])
ops = [e for e in evs if e.get("cat") == "cpu_op"]
cnt = collections.Counter()
for r in rt:
    t = r["ts"]
    best = None
    for e in py:
        if e["ts"] <= t <= e["ts"] + e["dur"] and (best is None or e["dur"] < best["dur"]):
            best = e
    op = None
    for e in ops:
        if e["ts"] <= t <= e["ts"] + e["dur"] and (op is None or e["dur"] < op["dur"]):
            op = e
    cnt[(r["name"], op["name"] if op else "?", best["name"].split("vspbfr_amd/")[-1] if best else "?")] += 1
for k, n in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{n:5d}  {k}")
os.remove(path)
