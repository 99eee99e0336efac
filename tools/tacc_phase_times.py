"""Phase times of the persistent sampler chain (tuning build -DVSP_TP_TIMING of tacc_persist.hip: cycles of workgroup 0 per phase).
usage: VSPBFR_HIP_LIB=build/abl/libvspbfr_tp.so python tools/tacc_phase_times.py [cluster] [B] [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd.diffusion import Code_diffuser, My_DDPM
from vspbfr_amd import hip_ops as H
cl = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = Code_diffuser(timesteps=T).to(dev).eval()
ddpm = My_DDPM(denoise=net, linear_start=0.1, linear_end=0.99, timesteps=T).to(dev)
cond = torch.randn(B, 18, 512, device=dev)
H.TACC_PERSISTENT, H.TACC_CLUSTER = True, cl
for _ in range(3):
    ddpm(x=cond, condi_in=cond, training=False)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); ddpm(x=cond, condi_in=cond, training=False); e.record(); torch.cuda.synchronize()
w = H._TACC_LAST_WORK
M = B * 18
off = M * 4 * 512 + M * 512 * 2 + 2 * M * 512
ph = w[off + 40: off + 45].view(torch.int32).cpu().tolist()
tot = sum(ph)
ms = s.elapsed_time(e)
names = ["proj", "barrier 1", "attention", "barrier 2", "post"]
print(f"cluster {cl}, B={B}, T={T}: {ms:.2f} ms; per block " + ", ".join(f"{n} {v / tot * ms * 1e3 / (4 * T):.1f} us" for n, v in zip(names, ph)))
