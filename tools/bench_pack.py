import torch, time, sys
sys.path.insert(0, '.')
from vspbfr_amd import hip_ops as H
for shape in [(512,512,3,3),(256,512,3,3),(64,3,7,7),(512,512,1,1),(128,128,3,3)]:
    w = torch.randn(*shape, device='cuda')
    for name, fn in (("pack", lambda: H.pack_weight(w, scale=0.5)), ("adjoint", lambda: H.pack_weight(w, adjoint=True, flip=True, scale=0.5))):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t=time.perf_counter()
        for _ in range(50): fn()
        torch.cuda.synchronize(); print(shape, name, round((time.perf_counter()-t)/50*1e6,1), "us")
    if shape[2]==3:
        wp = H.pack_weight(w)
        for _ in range(3): H.winograd_weight(wp)
        torch.cuda.synchronize(); t=time.perf_counter()
        for _ in range(50): H.winograd_weight(wp)
        torch.cuda.synchronize(); print(shape, "winograd", round((time.perf_counter()-t)/50*1e6,1), "us")
