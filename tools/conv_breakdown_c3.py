"""Per-shape conv kernel time of one serial step of BASELINE configs[2] (B = 16, bf16 kernels, bf16 activations in HBM): HIP events per
launch, with the algorithmic HBM bytes of each launch (input + output + residuals once, 2 B per activation element) and the rates."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vspbfr_amd import hip_ops
dev = torch.device("cuda", 0)
B, T = int(os.environ.get("B", 16)), 50
pipe = bench.build_pipeline(dev, T, True)
pipe.act_bf16 = True
hip_ops.BF16_CONV = True
lq = torch.rand(B, 3, 512, 512, device=dev) * 2 - 1
with torch.no_grad():
    pipe(lq); pipe(lq)
    prof = hip_ops.ConvProfiler(); hip_ops.PROFILER = prof
    pipe(lq)
    hip_ops.PROFILER = None
torch.cuda.synchronize()
agg = collections.OrderedDict()
for fl, s, e, tag, nb in prof.records:
    k = tag[:7]
    a = agg.setdefault(k, [0, 0.0, 0.0, 0.0, set()])
    a[0] += 1; a[1] += s.elapsed_time(e); a[2] += fl; a[3] += nb; a[4].add(tag[7])
tot = 0.0
print("Cin,Cout,OH,OW,k,stride,G | n | ms | TFLOP/s | algorithmic GB | TB/s | kernel")
for k in sorted(agg, key=lambda k: -agg[k][1]):
    n, ms, fl, nb, kinds = agg[k]
    tot += ms
    print(f"{k} | {n} | {ms:.2f} | {fl / ms / 1e9:.0f} | {nb / 1e9:.2f} | {nb / ms / 1e9:.2f} | {','.join(sorted(kinds))}")
print(f"total {tot:.2f} ms")
