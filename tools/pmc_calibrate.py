"""Kernels with exactly known HBM byte counts, in this library's own access patterns, to calibrate rocprofv3's
FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 for wide coalesced loads, other widths and
WRITE_SIZE are uncalibrated)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vspbfr_amd import hip_ops as H

dev = "cuda"
e = torch.empty(0, device=dev)
# 1 GiB in / 1 GiB out, far beyond the 256 MiB Infinity Cache
x = torch.randn(8, 64, 724, 724, device=dev)       # 16-byte loads/stores (n % 4 == 0, plane % 4 == 0)
b = torch.randn(64, device=dev)
for _ in range(2):
    H.fused_bias_act(x, b, e, 3, 0, 0.2, 1.4142)   # fba_vec4_kernel: reads 4N, writes 4N bytes
x2 = torch.randn(8, 64, 723, 723, device=dev)      # odd plane -> scalar kernel: 4-byte loads/stores
for _ in range(2):
    H.fused_bias_act(x2, b, e, 3, 0, 0.2, 1.4142)  # fba_scalar_kernel
torch.cuda.synchronize()
print("bytes_vec4", x.numel() * 4, "bytes_scalar", x2.numel() * 4)
