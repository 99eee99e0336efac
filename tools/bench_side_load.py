"""What a side stream of small kernels costs the big convolutions it runs under (round 6).  Stages C + D of one batch (26 ms alone at batch 8)
timed with a synthetic load on the second stream: N launches per batch of (a) an empty kernel of one workgroup, (b) an empty kernel of 256
workgroups x 512 threads (the sampler chain's launch shape), (c) the same shape streaming 4 MB of weights per launch through L2, (d) the same
shape spinning ~8 us, (e) the real chain.  usage: bench_side_load.py [--launches 600] [--steps 6]"""
import argparse
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--launches", type=int, default=600)
ap.add_argument("--steps", type=int, default=6)
args = ap.parse_args()
dev = torch.device("cuda", 0)
from vspbfr_amd import hip_ops, pipeline  # noqa: E402

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "side_load.so"))
lib.side_load.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
B = 8
pipe = bench.build_pipeline(dev, 50, True, 123)
lq = hip_ops.keyed_fill([(B, 3, 512, 512)], [hip_ops.SEG_LQ], 123, 0, dist="uniform", device=dev)[0]
weights = torch.randn(16 * 1024 * 1024 // 4, device=dev)   # 16 MB: the four TACC blocks' weights of one step
side = pipeline._side_stream()
main = torch.cuda.current_stream()
K, N = args.steps, args.launches

with torch.no_grad():
    lat, pre = pipe.encode(lq, image_index0=0)
    lat, pre = lat.clone(), pre.clone()

    def run(load):
        """K batches of C + D on the main stream, the load of batch i on the side stream from the moment C + D of batch i starts; returns the
        main stream's own time per batch (events on the main stream: the side stream's tail is not in it)."""
        def once(n):
            for i in range(n):
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                if load is not None:
                    with torch.cuda.stream(side):
                        load()
                pipe.decode(lq, lat, pre, image_index0=i * B)
        once(2)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        once(K)
        e1.record(main)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / K

    def L(wgs, threads, src=None, n16=0, spin=0):
        return lambda: lib.side_load(N, wgs, threads, src, n16, spin, None, ctypes.c_void_p(side.cuda_stream))

    def chain():
        pipe.diffusion(x=lat, condi_in=lat, training=False)

    def load_alone(load):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(side):
            for _ in range(K):
                load()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / K * 1e3

    wp = weights.data_ptr()
    cases = [("none", None), (f"{N} x empty 1 wg x 64", L(1, 64)), (f"{N} x empty 256 wg x 512", L(256, 512)),
             (f"{N} x 256 wg x 512 streaming 4 MB", L(256, 512, wp, 4 * 1024 * 1024 // 16 // 256, 0)),
             (f"{N} x 256 wg x 512 spinning", L(256, 512, None, 0, 2000)),
             (f"{N} x 32 wg x 512 streaming 4 MB", L(32, 512, wp, 4 * 1024 * 1024 // 16 // 32, 0)),
             (f"{N // 6} x 256 wg x 512 streaming 4 MB", lambda: lib.side_load(N // 6, 256, 512, wp, 4 * 1024 * 1024 // 16 // 256, 0, None, ctypes.c_void_p(side.cuda_stream))),
             ("the real chain (T = 50)", chain), ("none again", None)]
    for name, load in cases:
        alone = load_alone(load) if load is not None else 0.0
        print(f"{name:45s}: load alone {alone:6.2f} ms | C + D with it {run(load):6.2f} ms", flush=True)
