"""Timeline of the two-stream batch loop (round 6): when does the side stream's share of stages A + B (small-map head stages + sampler chain)
start and finish relative to the main stream's C + D?  If decode(i) starts right when side(i) ends, the side stream is the critical path
(its latency-bound launches wait for workgroup slots under the big convolutions) and the loop is bound by it, not by the chip.
usage: overlap_timeline.py [--preset c2|c3] [--split h|b|ab] [--steps K]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--preset", default="c2")
ap.add_argument("--split", default="h")
ap.add_argument("--steps", type=int, default=6)
args = ap.parse_args()
cfg = bench.PRESETS[args.preset]
B = cfg["batch"]
dev = torch.device("cuda", 0)
from vspbfr_amd import hip_ops, pipeline  # noqa: E402

pipe = bench.build_pipeline(dev, cfg["timesteps"], True, 123)
pipe.overlap_split = args.split
if cfg.get("sampler") == "ddim":
    from vspbfr_amd.ddim import DDIMSampler
    sampler, S = DDIMSampler(pipe.diffusion, device=dev), cfg["ddim_steps"]

    class _DDIM(torch.nn.Module):
        def forward(self, x=None, condi_in=None, training=False, x_T=None):
            return sampler.sample(S=S, batch_size=condi_in.shape[0], shape=18 * 512, conditioning=condi_in, eta=0.0, verbose=False, x_T=x_T)[0]
    pipe.diffusion = _DDIM()
hip_ops.BF16_CONV = cfg.get("conv_dtype") == "bf16"
pipe.act_bf16 = bool(cfg.get("act_bf16"))
lq = hip_ops.keyed_fill([(B, 3, 512, 512)], [hip_ops.SEG_LQ], 123, 0, dist="uniform", device=dev)[0]

marks = []   # (label, batch, event)


def mark(label, i):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(torch.cuda.current_stream())
    marks.append((label, i, ev))


# instrument: wrap encode / decode and the hand-over
orig_encode, orig_decode = pipe.encode, pipe.decode
counter = {"e": 0, "d": 0}


def encode(batch, x_T=None, image_index0=0, handoff=None):
    i = counter["e"]
    counter["e"] += 1
    mark("A_start", i)
    if handoff is not None:
        go = handoff.go

        def go2(tensors):
            mark("A_main_end", i)
            go(tensors)
            mark("side_start", i)
        handoff.go = go2
    out = orig_encode(batch, x_T=x_T, image_index0=image_index0, handoff=handoff)
    mark("side_end", i)
    return out


def decode(*a, **k):
    i = counter["d"]
    counter["d"] += 1
    mark("CD_start", i)
    out = orig_decode(*a, **k)
    mark("CD_end", i)
    return out


pipe.encode, pipe.decode = encode, decode
with torch.no_grad():
    for _ in pipe.run_batches([(lq, i * B) for i in range(3)]):
        pass
    torch.cuda.synchronize()
    marks.clear()
    counter.update(e=0, d=0)
    base = torch.cuda.Event(enable_timing=True)
    base.record()
    for _ in pipe.run_batches([(lq, i * B) for i in range(args.steps)]):
        pass
    torch.cuda.synchronize()
rows = {}
for label, i, ev in marks:
    rows.setdefault(i, {})[label] = base.elapsed_time(ev)
print(f"preset {args.preset} split {args.split}: times in ms since the loop start")
for i in sorted(rows):
    r = rows[i]
    print(i, " ".join(f"{k}={r[k]:.2f}" for k in ("A_start", "A_main_end", "side_start", "side_end", "CD_start", "CD_end") if k in r),
          f"| side {r['side_end'] - r.get('side_start', r['A_start']):.2f} ms, C+D {r['CD_end'] - r['CD_start']:.2f} ms, C+D waited for side: {r['CD_start'] >= r['side_end'] - 0.05}")
