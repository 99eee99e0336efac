"""Per-shape conv kernel time of one serial step (HIP events per launch), fp32 vs the bf16 configuration."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vspbfr_amd import hip_ops
dev = torch.device("cuda", 0)
B, T = int(os.environ.get("B", 8)), int(os.environ.get("T", 50))
pipe = bench.build_pipeline(dev, T, True)
lq = torch.rand(B, 3, 512, 512, device=dev) * 2 - 1
def step(bf):
    hip_ops.BF16_CONV = (os.environ.get("X3") and "x3" or True) if bf else False
    with torch.no_grad():
        pipe(lq); pipe(lq)
        prof = hip_ops.ConvProfiler(); hip_ops.PROFILER = prof
        pipe(lq)
        hip_ops.PROFILER = None
    torch.cuda.synchronize()
    agg = collections.OrderedDict()
    FRAC = {"wino": 16.0 / 36.0, "wino4": 0.25, "wino4f": 0.25}   # share of the direct-form multiplies a Winograd launch executes
    for fl, s, e, tag, _nb in prof.records:
        k = tag[:7]
        a = agg.setdefault(k, [0, 0.0, 0.0, set(), 0.0])
        a[0] += 1; a[1] += s.elapsed_time(e); a[2] += fl; a[3].add(tag[7]); a[4] += fl * FRAC.get(tag[7], 1.0)   # executed: per LAUNCH (a row may mix kinds)
    return agg
a32, a16 = step(False), step(True)
tot32 = tot16 = 0.0
print("Cin,Cout,OH,OW,k,stride,G | n | fp32 ms (effective TF / executed TF) kinds | bf16-config ms (TF) kinds")
print("# effective = algorithmic 2*Cin*9 FLOP per output / time; executed = what the MFMA pipe ran: Winograd F(2x2,3x3) layers execute 16/36 of it, F(4x4,3x3) layers (wino4) 36/144")
for k in sorted(a32, key=lambda k: -a32[k][1]):
    n, ms, fl, kinds, ex = a32[k]; n2, ms2, fl2, kinds2, _ = a16[k]
    tot32 += ms; tot16 += ms2
    print(f"{k} | {n} | {ms:.2f} ({fl/ms/1e9:.0f} / {ex/ms/1e9:.0f}) {','.join(sorted(kinds))} | {ms2:.2f} ({fl2/ms2/1e9:.0f}) {','.join(sorted(kinds2))}")
print(f"total fp32 {tot32:.2f} ms, bf16 config {tot16:.2f} ms")
