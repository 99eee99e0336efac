import sys, time, torch
sys.path.insert(0, '.')
from vspbfr_amd.discriminator import Discriminator
dev = torch.device('cuda')
torch.manual_seed(0)
D = Discriminator(512).to(dev)
a, b = torch.rand(4, 3, 512, 512, device=dev), torch.rand(4, 3, 512, 512, device=dev)
def two():
    for p in D.parameters(): p.grad = None
    (D(a).sum() + D(b).sum()).backward()
def one():
    for p in D.parameters(): p.grad = None
    D(torch.cat([a, b])).sum().backward()
for name, fn in (("two passes of 4", two), ("one pass of 8", one)):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); print(name, round((time.perf_counter() - t) / 5 * 1e3, 2), "ms")
