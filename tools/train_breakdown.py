"""Per-shape time of the convolution launches of ONE training iteration (forward, data-gradient and weight-gradient kernels; HIP
events per launch) and the share of everything else.  usage: python tools/train_breakdown.py [B] [losses]"""
import collections, copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vspbfr_amd import hip_ops
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
tr = RestorationTrainer(G, copy.deepcopy(G), D, psp_embedding=pipe.psp, diffusion=pipe.diffusion, mixing=0.9, **kw)
low, real = torch.rand(B, 3, 512, 512, device=dev) * 2 - 1, torch.rand(B, 3, 512, 512, device=dev) * 2 - 1
G.train()
for i in (1, 2):
    tr.step(i, low, real)
torch.cuda.synchronize()
t0 = time.perf_counter(); tr.step(3, low, real); torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
prof = hip_ops.ConvProfiler(); hip_ops.PROFILER = prof
tr.step(4, low, real)
hip_ops.PROFILER = None
torch.cuda.synchronize()
agg = collections.OrderedDict()
for fl, s, e, tag, _nb in prof.records:
    k = tag[:8]
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += s.elapsed_time(e); a[2] += fl
tot = sum(a[1] for a in agg.values())
kinds = collections.Counter()
print(f"iteration {wall:.1f} ms wall; conv launches {sum(a[0] for a in agg.values())}, {tot:.1f} ms")
print("Cin_g,Cout_g,OH,OW,k,stride,G,kind | n | ms | TFLOP/s")
for k in sorted(agg, key=lambda k: -agg[k][1]):
    n, ms, fl = agg[k]
    kinds[k[7]] += ms
    if ms > 0.4:
        print(f"{k} | {n} | {ms:.2f} | {fl / ms / 1e9:.1f}")
print({k: round(v, 1) for k, v in kinds.items()})
