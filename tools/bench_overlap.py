"""A/B: plain per-batch loop vs RestorationPipeline.run_batches (A+B of the next batch on a second stream)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
pipe = bench.build_pipeline(dev, 50, True)
lq = torch.rand(8, 3, 512, 512, device=dev) * 2 - 1
with torch.no_grad():
    pipe(lq); 
    for _ in pipe.run_batches([lq, lq]): pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): r = pipe(lq)["restored"]
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for o in pipe.run_batches([lq] * K): r2 = o["restored"]
    torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"plain {K} batches: {(t1-t0)/K*1e3:.2f} ms/batch ({8*K/(t1-t0):.1f} img/s) | pipelined: {(t2-t1)/K*1e3:.2f} ms/batch ({8*K/(t2-t1):.1f} img/s)")
print("finite", bool(torch.isfinite(r2).all()))
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range", lo, hi)
for pr_side, pr_main in ((0, -1), (-1, 0)):
    pipe._side = torch.cuda.Stream(priority=pr_side)
    mainS = torch.cuda.Stream(priority=pr_main)
    with torch.no_grad(), torch.cuda.stream(mainS):
        for _ in pipe.run_batches([lq, lq]): pass
        torch.cuda.synchronize(); t1 = time.perf_counter()
        for o in pipe.run_batches([lq] * K): r2 = o["restored"]
        torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"side prio {pr_side} main prio {pr_main}: {(t2-t1)/K*1e3:.2f} ms/batch ({8*K/(t2-t1):.1f} img/s)")

