"""Single-batch latency and host/GPU balance: wall time per batch vs the sum of kernel time (HIP events over the step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
for B, T in ((1, 4), (1, 50), (2, 4), (4, 4)):
    pipe = bench.build_pipeline(dev, T, True)
    lq = torch.rand(B, 3, 512, 512, device=dev) * 2 - 1
    with torch.no_grad():
        for _ in range(3): pipe(lq)
        torch.cuda.synchronize()
        N = 10
        t0 = time.perf_counter()
        for _ in range(N): pipe(lq)
        t_host = time.perf_counter() - t0      # time to ENQUEUE N steps
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
    print(f"B={B} T={T}: {t_all/N*1e3:.2f} ms per batch wall, host enqueue {t_host/N*1e3:.2f} ms per batch -> {B*N/t_all:.1f} img/s")
