"""The ATen ops of one training iteration ranked by the bytes they move (elements of their largest operand x 4): where the torch glue
between the kernels is worth replacing.  The call-site column is filled only where the profiler recorded a Python stack (it does not
on the autograd thread: "?").  usage: python tools/trace_train_ops.py [B] [losses]"""
import collections, copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
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
tr.step(1, low, real)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    tr.step(2, low, real)
    torch.cuda.synchronize()
SKIP = ("aten::empty", "aten::empty_like", "aten::empty_strided", "aten::view", "aten::as_strided", "aten::select", "aten::slice",
        "aten::unsqueeze", "aten::reshape", "aten::_unsafe_view", "aten::expand", "aten::alias", "aten::t", "aten::transpose",
        "aten::permute", "aten::squeeze", "aten::resize_", "aten::detach", "aten::item", "aten::_local_scalar_dense", "aten::to",
        "aten::lift_fresh", "aten::result_type", "aten::stride", "aten::is_nonzero", "aten::detach_", "aten::_to_copy", "aten::flatten",
        "aten::view_as", "aten::expand_as", "aten::contiguous", "aten::unflatten")
agg = collections.defaultdict(lambda: [0, 0])
byshape = collections.defaultdict(lambda: [0, 0])
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.name in SKIP:
        continue
    shapes = [tuple(s_) for s_ in (ev.input_shapes or []) if s_]
    n = max([int(torch.tensor(s_).prod()) for s_ in shapes] + [0])
    site = "?"
    for fr in (ev.stack or []):
        if "vspbfr_amd/" in fr:
            site = fr.split("vspbfr_amd/")[-1].split(",")[0]
            break
        if "torch/optim" in fr:
            site = "torch.optim"; break
    a = agg[(ev.name, site)]
    a[0] += 1; a[1] += n
    b = byshape[(ev.name, tuple(shapes))]
    b[0] += 1; b[1] += n
tot = sum(a[1] for a in agg.values())
print(f"ATen ops: {sum(a[0] for a in agg.values())}, {tot * 4 / 1e9:.2f} GB of largest operands")
for (name, site), (cnt, n) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{n * 4 / 1e6:9.1f} MB  x{cnt:4d}  {name:26s} {site}")
print("-- by launch count")
for (name, site), (cnt, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:40]:
    print(f"x{cnt:4d}  {n * 4 / 1e6:9.1f} MB  {name:26s} {site}")
print("-- by (op, operand shapes)")
for (name, shapes), (cnt, n) in sorted(byshape.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{n * 4 / 1e6:9.1f} MB  x{cnt:4d}  {name:22s} {shapes}")
