"""Phase split of one restoration_train iteration (HIP events): frozen front, fake batch for D, D forward/backward (+Adam), G forward,
D(fake) + losses, G backward (+all-reduce), Adam + EMA.  usage: python tools/train_phases.py [B] [losses]"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vspbfr_amd.discriminator import Discriminator, accumulate, d_logistic_loss, g_nonsaturating_loss
from vspbfr_amd.restorenet import mixing_noise
from vspbfr_amd.train_step import RestorationTrainer, requires_grad

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
marks = []
def mark(name):
    e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e))
mark("start")
de_feats, latent = tr.front(low); de_feats = [f.detach() for f in de_feats]; mark("frozen front (A, B, C)")
requires_grad(G, False); requires_grad(D, True)
with torch.no_grad():
    fake = tr.generate(low, de_feats, latent, mixing_noise(B, G.style_dim, 0.9, dev))
mark("fake batch for D (fused forward)")
fp, rp = D(fake.detach()), D(real)
d_loss = d_logistic_loss(rp, fp); mark("D forward x2")
D.zero_grad(set_to_none=True); d_loss.backward(); mark("D backward")
tr.d_optim.step(); mark("D Adam")
requires_grad(G, True); requires_grad(D, False)
fake = tr.generate(low, de_feats, latent, mixing_noise(B, G.style_dim, 0.9, dev)); mark("G forward (differentiable)")
g_loss = g_nonsaturating_loss(D(fake)); mark("D(fake) forward")
if kw:
    g_loss = g_loss + tr.percept_loss(fake, real).sum() * 0.5; mark("LPIPS forward")
    g_loss = g_loss + tr.id_loss(fake, real) * 0.1; mark("ID forward")
G.zero_grad(set_to_none=True); g_loss.backward(); mark("backward (losses, D, G)")
tr.g_optim.step(); accumulate(tr.G_ema, G, tr.accum); mark("G Adam + EMA")
torch.cuda.synchronize()
tot = marks[0][1].elapsed_time(marks[-1][1])
for (n0, e0), (n1, e1) in zip(marks, marks[1:]):
    print(f"{n1:36s} {e0.elapsed_time(e1):7.1f} ms")
print(f"{'total':36s} {tot:7.1f} ms")
