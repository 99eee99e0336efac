"""Experiment: one restoration_train iteration captured as ONE HIP graph (forward, backward, both Adam steps, EMA) and replayed.
Fixed structure for the capture (single mixing code, no R1 pass); answers how much of the iteration is launch / dependency gaps of
the ~7 000 small torch ops and ~1 500 kernel launches.  usage: python tools/bench_train_graph.py [B]"""
import copy, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from vspbfr_amd.discriminator import Discriminator
from vspbfr_amd.train_step import RestorationTrainer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
pipe = bench.build_pipeline(dev, 4, False)
G = pipe.generator
torch.manual_seed(1)
D = Discriminator(512).to(dev)
tr = RestorationTrainer(G, copy.deepcopy(G), D, psp_embedding=pipe.psp, diffusion=pipe.diffusion, mixing=0.0)
for opt in (tr.g_optim, tr.d_optim):      # the step counter must live on the device for a captured optimiser step
    opt.defaults["capturable"] = True
    for grp in opt.param_groups:
        grp["capturable"] = True
low, real = torch.rand(B, 3, 512, 512, device=dev) * 2 - 1, torch.rand(B, 3, 512, 512, device=dev) * 2 - 1
G.train()

def eager(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        tr.step(1, low, real)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        tr.step(1, low, real)
torch.cuda.current_stream().wait_stream(side)
ms_eager = eager(3)
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    losses = tr.step(1, low, real)
torch.cuda.synchronize()
for _ in range(2):
    graph.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    graph.replay()
torch.cuda.synchronize(); ms_graph = (time.perf_counter() - t0) / 5 * 1e3
print(json.dumps({"what": "restoration_train iteration, batch %d, eager vs one captured HIP graph" % B, "ms_eager": round(ms_eager, 1),
                  "ms_graph_replay": round(ms_graph, 1), "losses": {k: float(v) for k, v in losses.items() if torch.is_tensor(v) and v.numel() == 1}}))
