"""Stage A alone (the e4e encoder of one batch of 8), K times: for rocprofv3 kernel traces.  usage: python tools/run_encoder.py [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pipe = bench.build_pipeline(dev, 50, True)
lq = torch.rand(8, 3, 512, 512, device=dev) * 2 - 1
with torch.no_grad():
    for _ in range(2):
        pipe.psp.get_w_plus(lq)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        pipe.psp.get_w_plus(lq)
    torch.cuda.synchronize()
print(f"encoder: {(time.perf_counter() - t0) / K * 1e3:.2f} ms per batch of 8")
