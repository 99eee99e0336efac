import sys; sys.path.insert(0, "/root/repo")
import torch, bench
from vspbfr_amd import hip_ops
dev = torch.device("cuda", 0)
pipe = bench.build_pipeline(dev, 4, True)
for B in (1, 3, 5):
    lq = torch.rand(B, 3, 512, 512, device=dev) * 2 - 1
    ref = None
    for mode in (False, "x3", True):
        hip_ops.BF16_CONV = mode
        torch.manual_seed(3)
        with torch.no_grad():
            o = pipe(lq)
        torch.cuda.synchronize()
        r = o["restored"]
        assert torch.isfinite(r).all()
        if ref is None: ref = r
        print(B, mode, float(r.std()), float((r - ref).abs().max()))
