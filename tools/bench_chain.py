"""Time the sampler chain alone (B=8, T=50).  usage: bench_chain.py [B] [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd.diffusion import Code_diffuser, My_DDPM
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = Code_diffuser(timesteps=T).to(dev).eval()
ddpm = My_DDPM(denoise=net, linear_start=0.1, linear_end=0.99, timesteps=T).to(dev)
cond = torch.randn(B, 18, 512, device=dev)
for _ in range(2):
    ddpm(x=cond, condi_in=cond, training=False)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
N = 5
for _ in range(N):
    ddpm(x=cond, condi_in=cond, training=False)
e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e) / N
print(f"chain B={B} T={T} (three launches per TACC block): {ms:.3f} ms  ({ms * 1000 / (4 * T):.1f} us per TACC block incl. prepare)")
