"""Autotune the conv tile configuration per layer shape of the restoration path (GPU box).

Runs the pipeline once (batch B, T steps) recording every vsp_conv2d_f32 launch, then times every compiled tile
configuration on each distinct shape and writes the winners to gpurun_out/conv_tune.json (copy to
vspbfr_amd/conv_tune.json to ship it).  usage: python tools/autotune_conv.py [B] [T]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from vspbfr_amd import hip_ops as H
from vspbfr_amd._lib import lib


WINO_ID = 1 << 20  # pseudo configuration id of the Winograd kernel in the timing tables


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    dev = torch.device("cuda", 0)
    H.TUNE, H.WINO = {}, {}
    pipe = bench.build_pipeline(dev, T, True)
    lq = torch.rand(B, 3, 512, 512, device=dev) * 2 - 1
    H.RECORDER = []
    with torch.no_grad():
        pipe(lq)
    torch.cuda.synchronize()
    recs, H.RECORDER = H.RECORDER, None
    uniq = {}
    for key, dims, pc, tr in recs:
        uniq.setdefault(key, [dims, pc, 0, tr])[2] += 1
    print(f"{len(recs)} conv launches, {len(uniq)} distinct shapes", flush=True)
    n = lib.vsp_conv2d_num_configs()
    table, report, cands = {}, [], {}
    tot_auto = tot_best = 0.0
    for key, (dims, pc, count, tr) in uniq.items():
        Bq, Cin, Hh, Ww, OH, OW = dims
        x = torch.randn(Bq, (pc.G - 1) * pc.x_group_stride + Cin, Hh, Ww, device=dev)
        shift = torch.zeros(Cin, device=dev)
        # generous output tensor: phase convs write strided, give them room
        out = torch.empty(Bq, pc.cout, max(OH, 1) * 2 + 1, max(OW, 1) * 2 + 1, device=dev)
        flops = 2.0 * Bq * pc.cout * (Hh * Ww if tr else OH * OW) * Cin * pc.kh * pc.kw
        times = {}
        for c in range(0, n + 1):
            try:
                kw = dict(transposed=True) if tr else dict(out=out, n_out=(OH, OW))
                if key.endswith(",s"):  # input shift (folded BatchNorm): not every staging variant serves it
                    kw["in_shift"] = shift
                est = timeit(lambda: H.conv2d_packed(x, pc, tile_hint=c, **kw), 1)
                iters = 3 if est > 0.3 else 10
                times[c] = timeit(lambda: H.conv2d_packed(x, pc, tile_hint=c, **kw), iters)
            except RuntimeError:
                continue
        if not [c for c in times if c > 0]:
            print("no configuration ran for", key, flush=True)
            continue
        name_of = lambda c: "winograd" if c == WINO_ID else lib.vsp_conv2d_config_name(c - 1).decode()  # noqa: E731
        if not tr and H.winograd_eligible(pc, Hh, Ww, OH, OW):
            try:
                kw = dict(out=out, n_out=(OH, OW), winograd=True)
                if key.endswith(",s"):
                    kw["in_shift"] = shift
                timeit(lambda: H.conv2d_packed(x, pc, **kw), 1)
                times[WINO_ID] = timeit(lambda: H.conv2d_packed(x, pc, **kw), 5)
            except RuntimeError:
                pass
        best = min((c for c in times if c > 0), key=lambda c: times[c])
        table[key] = name_of(best)
        order = sorted((c for c in times if c > 0), key=lambda c: times[c])[:5]
        cands[key] = [name_of(c) for c in order]
        tot_auto += times[0] * count
        tot_best += times[best] * count
        report.append((times[best] * count, key, count, name_of(best), round(times[best] * 1e3, 1),
                       round(times[0] * 1e3, 1), round(flops / times[best] / 1e9, 1)))
    report.sort(reverse=True)
    for r in report[:60]:
        print("%.2f ms total | %s | x%d | best %s %s us (cost-model pick %s us) | %s TF" % r, flush=True)
    print(f"sum per pipeline pass: cost model {tot_auto:.1f} ms -> tuned {tot_best:.1f} ms")
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(table, open("gpurun_out/conv_tune.json", "w"), indent=0, sort_keys=True)
    json.dump(cands, open("gpurun_out/conv_candidates.json", "w"), indent=0, sort_keys=True)
    insitu(pipe, lq, cands, table)


def insitu(pipe, lq, cands, table):
    """Second stage: the micro-benchmark replays bare launches with warm caches; in the pipeline a layer also loads its
    epilogue operands and meets whatever the previous kernel left in L2.  Run the pipeline once per candidate rank (every
    shape uses its rank-j candidate), time each launch with HIP events, keep the per-shape winner."""
    by_key = {}
    for j in range(5):
        pick = {k: v[min(j, len(v) - 1)] for k, v in cands.items()}
        H.TUNE = {k: H.CONFIG_IDS[v] for k, v in pick.items() if v != "winograd"}
        H.WINO = {k: True for k, v in pick.items() if v == "winograd"}
        for rep in range(2):
            prof = H.ConvProfiler()
            H.PROFILER = prof
            H.RECORDER = []
            with torch.no_grad():
                pipe(lq)
            torch.cuda.synchronize()
            H.PROFILER, keys, H.RECORDER = None, H.RECORDER, None
        for (key, _, _, _), rec in zip(keys, prof.records):
            by_key.setdefault(key, [0.0] * 5)[j] += rec[1].elapsed_time(rec[2])
    final, t_micro, t_best = {}, 0.0, 0.0
    for key, ts in by_key.items():
        n = len(cands[key])
        jbest = min(range(n), key=lambda j: ts[j])
        final[key] = cands[key][jbest]
        t_micro += ts[0]
        t_best += ts[jbest]
        if jbest != 0 and ts[0] - ts[jbest] > 0.02:
            print(f"in situ: {key}: {cands[key][0]} {ts[0]:.3f} ms -> {cands[key][jbest]} {ts[jbest]:.3f} ms", flush=True)
    for key in table:
        final.setdefault(key, table[key])
    print(f"in-situ conv time per pass: micro-benchmark winners {t_micro:.1f} ms -> in-situ winners {t_best:.1f} ms")
    json.dump(final, open("gpurun_out/conv_tune.json", "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    main()
