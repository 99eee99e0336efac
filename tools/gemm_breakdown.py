"""Round 3: the small-GEMM launches of one inference step by shape and call site (GPU box): count, total HIP-event time."""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vspbfr_amd import hip_ops as H

dev = torch.device("cuda", 0)
pipe = bench.build_pipeline(dev, 50, True, noise_seed=123)
lq = torch.rand(8, 3, 512, 512, device=dev) * 2 - 1
with torch.no_grad():
    pipe(lq); pipe(lq)
torch.cuda.synchronize()
recs = []
orig = H.gemm_nt


def hooked(a, b, *args, **kw):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    out = orig(a, b, *args, **kw)
    e.record()
    site = "?"
    for fr in reversed(traceback.extract_stack()[:-1]):
        if "vspbfr_amd" in fr.filename and "hip_ops" not in fr.filename:
            site = f"{os.path.basename(fr.filename)}:{fr.lineno}"
            break
    recs.append((tuple(a.shape), tuple(b.shape), site, s, e))
    return out


H.gemm_nt = hooked
import vspbfr_amd.layers, vspbfr_amd.e4e, vspbfr_amd.diffusion, vspbfr_amd.restorenet  # noqa: E401  (modules call H.gemm_nt through H)
with torch.no_grad():
    pipe(lq)
torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for a, b, site, s, e in recs:
    k = (site, a, b)
    agg[k][0] += 1
    agg[k][1] += s.elapsed_time(e)
print(f"{len(recs)} gemm launches, {sum(v[1] for v in agg.values()):.2f} ms (event intervals, serial step)")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{v[0]:4d} x  {v[1]*1e3/v[0]:7.1f} us  total {v[1]:.3f} ms   {k}")
