import os, sys
sys.path.insert(0, '.')
import numpy as np, torch
from oracle import cases, weights
from vspbfr_amd import hip_ops as H
from vspbfr_amd.id_loss import IDLoss
g = dict(np.load('tests/golden/idloss128.npz'))
sd = weights.synth_state_dict("arcface_resnet101", weights.load_specs()["arcface_resnet101"], cases.SEED)
sd.update({k[3:]: torch.from_numpy(g[k]) for k in g if k.startswith("bn/")})
DEV = torch.device('cuda')
pred, target = cases.image_batch("idloss128/pred", 2, 128).to(DEV), cases.image_batch("idloss128/target", 2, 128).to(DEV)
ref = g["d_pred"]
SM = H.CONFIG_IDS["smallmap"]
tune0 = dict(H.TUNE)
def run(tag):
    idl = IDLoss(sd, device=DEV)
    x = pred.clone().requires_grad_(True)
    loss = idl(x, target); (loss * 0.1).backward()
    d = x.grad.cpu().numpy()
    print(tag, "rel L2 %.3e  max %.3e of %.3e  loss err %.2e" % (np.linalg.norm(d - ref) / np.linalg.norm(ref), np.abs(d - ref).max(), np.abs(ref).max(), abs(loss.item() - float(g["loss"][0]))))
    return d
a = run("all new kernels      ")
H.CONV1X1_SMALL_MAX_P = 0
b = run("no 1x1 gemm          ")
H.TUNE.clear(); H.TUNE.update({k: v for k, v in tune0.items() if v != SM})
c = run("no 1x1 gemm, no smallmap in table")
H.CONV1X1_SMALL_MAX_P = 1024
d = run("1x1 gemm only        ")
print("a vs c rel", np.linalg.norm(a - c) / np.linalg.norm(c))
