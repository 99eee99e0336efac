"""Host/GPU balance: wall time per batch vs the host time to ENQUEUE a batch (plain serial loop), fp32 and bf16 configurations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vspbfr_amd import hip_ops
dev = torch.device("cuda", 0)
cfgs = [(8, 50), (1, 4), (16, 50)] if len(sys.argv) < 2 else [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for B, T in cfgs:
    pipe = bench.build_pipeline(dev, T, True)
    lq = torch.rand(B, 3, 512, 512, device=dev) * 2 - 1
    for bf in (False, True):
        hip_ops.BF16_CONV = bf
        with torch.no_grad():
            for _ in range(3): pipe(lq)
            torch.cuda.synchronize()
            N = 6
            t0 = time.perf_counter()
            for _ in range(N): pipe(lq)
            t_host = time.perf_counter() - t0      # time to ENQUEUE N steps
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
        print(f"B={B} T={T} {'bf16' if bf else 'fp32'}: {t_all/N*1e3:.2f} ms per batch wall, host enqueue {t_host/N*1e3:.2f} ms per batch -> {B*N/t_all:.1f} img/s", flush=True)
    if os.environ.get("GRAPHS"):
        for bf in (False, True):
            hip_ops.BF16_CONV = bf
            p2 = bench.build_pipeline(dev, T, True)
            with torch.no_grad():
                p2.capture_graphs(lq)
                for _ in range(2):
                    for o in p2.run_batches_graphed([lq]): pass
                torch.cuda.synchronize()
                N = 6
                t0 = time.perf_counter()
                for _ in range(N):
                    for o in p2.run_batches_graphed([lq]): pass
                    torch.cuda.synchronize()     # latency: one batch at a time
                t_all = time.perf_counter() - t0
            print(f"B={B} T={T} {'bf16' if bf else 'fp32'} graphs, one batch at a time: {t_all/N*1e3:.2f} ms per batch -> {B*N/t_all:.1f} img/s", flush=True)
