"""Bit comparison of the sampler chain forms (launched / persistent clusters), repeated runs.  usage: python tools/chain_bits.py [B] [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd.diffusion import Code_diffuser, My_DDPM
from vspbfr_amd import hip_ops as H
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
torch.manual_seed(0)
net = Code_diffuser(timesteps=T).to(dev).eval()
ddpm = My_DDPM(denoise=net, linear_start=0.1, linear_end=0.99, timesteps=T).to(dev)
cond = torch.randn(B, 18, 512, device=dev)
xT = torch.randn(B, 18, 512, device=dev)
outs = {}
for pers, cl in ((False, 16), (True, 16), (True, 16), (True, 4), (True, 4), (True, 1)):
    H.TACC_PERSISTENT, H.TACC_CLUSTER = pers, cl
    y = ddpm(x=cond, condi_in=cond, training=False, x_T=xT.clone())
    key = ("launched" if not pers else f"cluster{cl}")
    outs.setdefault(key, []).append(y.clone())
ref = outs["launched"][0]
for k, v in outs.items():
    print(k, [float((t - ref).abs().max()) for t in v], "finite", all(bool(torch.isfinite(t).all()) for t in v))
