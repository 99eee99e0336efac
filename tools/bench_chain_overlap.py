"""What the sampler chain costs the batch loop: stages C + D of one batch on the caller's stream, alone and with ONLY the chain of the
next batch on a second stream (round 3 also timed persistent clusters here: branch experiments/persistent-chain).
usage: python tools/bench_chain_overlap.py [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vspbfr_amd import hip_ops as H
dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
pipe = bench.build_pipeline(dev, 50, True)
lq = torch.rand(8, 3, 512, 512, device=dev) * 2 - 1
side = torch.cuda.Stream()
with torch.no_grad():
    lat, pre = pipe.encode(lq)
    pipe.decode(lq, lat, pre)
    torch.cuda.synchronize()

    def run(chain):
        main = torch.cuda.current_stream()
        for _ in range(2 + K):
            if _ == 2:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            if chain is not None:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    pipe.diffusion(x=lat, condi_in=lat, training=False)
            pipe.decode(lq, lat, pre)
            if chain is not None:
                main.wait_stream(side)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e3

    base = run(None)
    print(f"C + D alone: {base:.2f} ms")
    t = run(True)
    print(f"+ chain (launched): {t:.2f} ms (+{t - base:.2f})")
