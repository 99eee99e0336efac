"""In-situ choice of the vsp_conv2d_bf16 tile variant per layer shape: the pipeline runs once per candidate variant with that
variant forced on every launch it serves (hip_ops.BF16_FORCE), per-launch HIP events give the time of every shape, the fastest
variant per shape goes to gpurun_out/conv_tune_bf16.json (copy to vspbfr_amd/conv_tune_bf16.json to ship it).
usage: [ACT_BF16=1] [X3=1] python tools/autotune_bf16.py [B] [T]"""
import collections, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vspbfr_amd import hip_ops

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
X3 = bool(os.environ.get("X3"))   # X3=1: tune the split-precision form (vsp_conv2d_bf16x3) -> conv_tune_bf16x3.json
hip_ops.BF16_CONV = "x3" if X3 else True
hip_ops.BF16_TUNE = {}
hip_ops.BF16X3_TUNE = {}
pipe = bench.build_pipeline(dev, T, True)
pipe.act_bf16 = bool(os.environ.get("ACT_BF16"))   # ACT_BF16=1: tune with bf16 activations in HBM (BASELINE configs[2])
lq = torch.rand(B, 3, 512, 512, device=dev) * 2 - 1


def measure(force, reps=3):
    hip_ops.BF16_FORCE = force
    agg = collections.defaultdict(float)
    with torch.no_grad():
        pipe(lq); pipe(lq)
        for _ in range(reps):
            prof = hip_ops.ConvProfiler(); hip_ops.PROFILER = prof
            pipe(lq)
            hip_ops.PROFILER = None
            torch.cuda.synchronize()
            for fl, s, e, tag, _nb in prof.records:
                if tag[7] in ("bf16", "bf16x3"):
                    agg[tag[8]] += s.elapsed_time(e) / reps
    hip_ops.BF16_FORCE = 0
    return agg


res = {v: measure(v) for v in ((0, 1, 4, 5, 6, 7) if X3 else (0, 1, 2, 3, 4, 5, 6, 7, 8))}
best, tot_auto, tot_best = {}, 0.0, 0.0
for key in sorted(res[0], key=lambda k: -res[0][k]):
    f = key.split(",")
    valid = {4, 5, 8} if key.endswith(",t") else ({4, 6} if f[8] == "2" else {1, 2, 3, 4, 6, 7})  # variants of the launch's mode
    if X3:
        valid = {5} if key.endswith(",t") else ({6, 7} if f[8] == "2" else {1, 4, 7})
    cand = {v: r[key] for v, r in res.items() if key in r and (v == 0 or v in valid)}  # (others fell back to the library rule)
    v = min(cand, key=cand.get)
    tot_auto += cand[0]; tot_best += cand[v]
    if v != 0 and cand[v] < 0.97 * cand[0]:
        best[key] = v
    print(f"{cand[0]:.3f} ms auto | {key} | best v{v} {cand[v]:.3f} ms | " + " ".join(f"v{k}:{t:.3f}" for k, t in sorted(cand.items())), flush=True)
print(f"bf16 conv time per step: library rule {tot_auto:.2f} ms -> per-shape best {tot_best:.2f} ms ({len(best)} shapes overridden)")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(best, open("gpurun_out/conv_tune_bf16x3.json" if X3 else "gpurun_out/conv_tune_bf16.json", "w"), indent=1, sort_keys=True)
