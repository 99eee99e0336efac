"""One iteration of the stage-B training loop (code_diffuser_train.py:153-190) at the real size: e4e encoder on the degraded and the
clean image, training-mode sampler (T = 4), the 1024^2 StyleGAN2 prior on the predicted codes pooled to `size`, LPIPS-VGG + ArcFace
terms, Adam on the 72 Code_diffuser tensors.  Random-init networks (no checkpoints here).
usage: python tools/bench_diffuser_train.py [B] [iters] [size]   -> JSON line"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vspbfr_amd.id_loss import IDLoss
from vspbfr_amd.lpips import PerceptualLoss
from vspbfr_amd.train_step import CodeDiffuserTrainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
size = int(sys.argv[3]) if len(sys.argv) > 3 else 256            # code_diffuser_train.py --size default
dev = torch.device("cuda", 0)
pipe = bench.build_pipeline(dev, 4, False)
pipe.psp.E4Enet.out_size = size
tr = CodeDiffuserTrainer(pipe.diffusion, pipe.psp, percept_loss=PerceptualLoss().to(dev), id_loss=IDLoss(None, device=dev))
low, real = torch.rand(B, 3, size, size, device=dev) * 2 - 1, torch.rand(B, 3, size, size, device=dev) * 2 - 1
times = []
for i in range(iters + 1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    losses = tr.step(low, real)
    torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
ms = sum(times[1:]) / iters * 1e3
print(json.dumps({"what": "code_diffuser_train step: e4e codes of both images, DDPM T=4 training forward, 1024^2 prior -> %d^2, LPIPS + ID, Adam" % size,
                  "batch_per_gpu": B, "ms_per_iteration": round(ms, 1), "img_per_s": round(B / ms * 1e3, 2),
                  "losses": {k: float(v) for k, v in losses.items() if v.numel() == 1},
                  "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1)}))
