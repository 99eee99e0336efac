"""What the sampler chain costs the batch loop: stages C + D of one batch on the caller's stream, alone and with ONLY the chain of the
next batch on a second stream, in its launched form and as persistent clusters of G workgroups per image.
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
                    H.TACC_PERSISTENT, H.TACC_CLUSTER = chain
                    pipe.diffusion(x=lat, condi_in=lat, training=False)
            pipe.decode(lq, lat, pre)
            if chain is not None:
                main.wait_stream(side)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e3

    base = run(None)
    print(f"C + D alone: {base:.2f} ms")
    for chain in ((False, 16), (True, 16), (True, 8), (True, 4), (True, 2)):
        t = run(chain)
        print(f"+ chain {'launched' if not chain[0] else 'cluster %d' % chain[1]}: {t:.2f} ms (+{t - base:.2f})")
