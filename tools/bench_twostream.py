"""Experiment: two whole-pipeline instances on two streams (alternating batches) vs run_batches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 6
pipe = bench.build_pipeline(dev, 50, True)
lq = torch.rand(8, 3, 512, 512, device=dev) * 2 - 1
ss = [torch.cuda.Stream(), torch.cuda.Stream()]
def two(n):
    outs = []
    for i in range(n):
        s = ss[i & 1]
        with torch.cuda.stream(s):
            outs.append(pipe(lq)["restored"])
    for s in ss: torch.cuda.current_stream().wait_stream(s)
    return outs
with torch.no_grad():
    pipe(lq); two(2); list(pipe.run_batches([lq, lq])); torch.cuda.synchronize()
    t0 = time.perf_counter(); [pipe(lq) for _ in range(K)]; torch.cuda.synchronize(); t1 = time.perf_counter()
    list(pipe.run_batches([lq] * K)); torch.cuda.synchronize(); t2 = time.perf_counter()
    two(K); torch.cuda.synchronize(); t3 = time.perf_counter()
print(f"K={K}: plain {(t1-t0)/K*1e3:.2f} | run_batches {(t2-t1)/K*1e3:.2f} | two full streams {(t3-t2)/K*1e3:.2f} ms/batch")
