"""List the ATen ops (not our C ABI launches) the pipeline still issues per batch, with their Python call sites."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
pipe = bench.build_pipeline(dev, int(os.environ.get("T", 50)), True, noise_seed=None if os.environ.get("TORCH_RNG") else 123)
lq = torch.rand(int(os.environ.get("B", 8)), 3, 512, 512, device=dev) * 2 - 1
with torch.no_grad():
    pipe(lq)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    with torch.no_grad():
        pipe(lq)
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::"):
        continue
    if ev.name in ("aten::empty", "aten::empty_like", "aten::empty_strided", "aten::view", "aten::as_strided", "aten::select",
                   "aten::slice", "aten::unsqueeze", "aten::reshape", "aten::_unsafe_view", "aten::expand", "aten::alias",
                   "aten::t", "aten::transpose", "aten::permute", "aten::squeeze", "aten::resize_", "aten::detach", "aten::item",
                   "aten::_local_scalar_dense", "aten::to", "aten::lift_fresh", "aten::result_type", "aten::stride", "aten::is_nonzero"):
        continue
    site = "?"
    for fr in (ev.stack or []):
        if "vspbfr_amd" in fr or "bench.py" in fr:
            site = "vspbfr_amd/" + fr.split("vspbfr_amd/")[-1] if "vspbfr_amd/" in fr else fr
            break
    shp = str([tuple(s_) for s_ in (ev.input_shapes or []) if s_][:2])
    cnt[(ev.name, site + " " + shp)] += 1
print("ATen ops per batch (excluding views / allocations):", sum(cnt.values()))
for (name, site), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:80]:
    print(f"{n:5d}  {name:28s} {site}")
