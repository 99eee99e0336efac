"""One training iteration (restoration_train.py:153-255) at the real size: Restoration_net(512) + Discriminator(512), batch B per GPU,
frozen front (stages A, B, C) through the inference kernels; with `losses` also the LPIPS-VGG (x 0.5) and ArcFace identity (x 0.1)
terms of BASELINE configs[4] (random-init VGG16 / ResNet-101: there is no network for checkpoints).
usage: python tools/bench_train_step.py [B] [iters] [losses]   -> JSON line with ms per iteration."""
import copy, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vspbfr_amd.discriminator import Discriminator
from vspbfr_amd.train_step import RestorationTrainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
pipe = bench.build_pipeline(dev, 4, False)                 # T = 4 sampler as restoration_train.py's load_ddpm default
G = pipe.generator
torch.manual_seed(1)
D = Discriminator(512).to(dev)
G_ema = copy.deepcopy(G)
LOSSES = len(sys.argv) > 3 and sys.argv[3] == "losses"
if "x3" in sys.argv[3:]:   # experiment: forward / data-gradient convolutions on the split-precision bf16 pipe (fp32-grade), weight gradients fp32
    from vspbfr_amd import hip_ops
    hip_ops.BF16_CONV = "x3"
kw = {}
if LOSSES:
    from vspbfr_amd.id_loss import IDLoss
    from vspbfr_amd.lpips import PerceptualLoss
    kw = dict(percept_loss=PerceptualLoss(model="net-lin", net="vgg").to(dev), percept_weight=0.5, id_loss=IDLoss(None, device=dev),
              id_weight=0.1)
tr = RestorationTrainer(G, G_ema, D, psp_embedding=pipe.psp, diffusion=pipe.diffusion, mixing=0.9, **kw)
low = torch.rand(B, 3, 512, 512, device=dev) * 2 - 1
real = torch.rand(B, 3, 512, 512, device=dev) * 2 - 1
G.train()
times = []
for i in range(iters + 1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    losses = tr.step(i + 1, low, real)                       # i + 1: no R1 step (every 16th iteration carries one)
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
# roofline leg: every convolution launch of one more iteration (forward, data gradient, weight gradient; loss networks included) with
# HIP events on the launch stream: algorithmic FLOPs / time against the fp32 MFMA peak (the path is fp32-compute bound)
from vspbfr_amd import hip_ops
prof = hip_ops.ConvProfiler(); hip_ops.PROFILER = prof
tr.step(iters + 2, low, real)
hip_ops.PROFILER = None
torch.cuda.synchronize()
fl, conv_ms, n_launch = prof.summary()
by_kind = {}
for f_, s_, e_, tag, _ in prof.records:
    k = "weight gradient" if tag[7] == "wgrad" else "forward + data gradient"
    a = by_kind.setdefault(k, [0.0, 0.0]); a[0] += f_; a[1] += s_.elapsed_time(e_)
roofline = {"bound": "mfma", "achieved": round(fl / conv_ms / 1e9, 1), "peak": 157.3, "unit": "TFLOP/s", "frac": round(fl / conv_ms / 1e9 / 157.3, 3),
            "traffic": None, "kernel": "conv family of one iteration: conv_wino / conv_igemm (forward, data gradient) + conv_wgrad",
            "launches": n_launch, "kernel_ms": round(conv_ms, 1), "algorithmic_gflop": round(fl / 1e9, 1),
            "split": {k: {"tflops": round(v[0] / v[1] / 1e9, 1), "ms": round(v[1], 1)} for k, v in by_kind.items()},
            "measured": "HIP events per launch on the launch stream over one iteration (serialises nothing: one stream)"}
torch.cuda.synchronize()
t0 = time.perf_counter()
tr.step(16, low, real)                                       # an iteration with the R1 regulariser (double backward)
torch.cuda.synchronize()
t_r1 = time.perf_counter() - t0
ms = sum(times[1:]) / iters * 1e3
print(json.dumps({"what": "restoration_train step, 512x512, Restoration_net + Discriminator fwd/bwd + Adam + EMA, frozen front, "
                          + ("LPIPS-VGG x0.5 + ArcFace ID x0.1" if LOSSES else "no LPIPS/ID"),
                  "batch_per_gpu": B, "ms_per_iteration": round(ms, 1), "img_per_s": round(B / ms * 1e3, 2),
                  "ms_iteration_with_r1": round(t_r1 * 1e3, 1), "losses": {k: float(v) for k, v in losses.items()},
                  "roofline": roofline, "dtype": "f32",
                  "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
                  "generator_param_mb": round(tr.generator_bytes / 2 ** 20, 1)}))
