"""Multi-process checks of the data-parallel path (gloo, world_size 2, CPU):
  * the batch split covers every image once, ragged splits work, and the all-gather reassembles the restored batch in rank
    order on every rank -- the same code path bench.py / the CLI use with RCCL on the GPUs;
  * a restoration-shaped computation (the oracle's size-64 Restoration_net standing in for the HIP kernels, which need a GPU)
    whose inputs and noise maps are keyed by the GLOBAL image index (oracle/device_rng.py = the numpy restatement of
    vsp_keyed_fill_f32) gives the same images rank-split as on one rank;
  * `python bench.py --gpus 2` launches itself (child torchrun, no GPU touched by the parent) and relays rank 0's JSON line."""
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, torch
    import torch.distributed as dist
    sys.path.insert(0, %r)
    from vspbfr_amd.pipeline import shard_range, gather_restored
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    for n in (8, 5):                       # even and ragged global batch
        full = torch.arange(n * 3 * 4 * 4, dtype=torch.float32).view(n, 3, 4, 4)
        lo, hi = shard_range(n, rank, world)
        local = full[lo:hi] * 2.0 + 1.0     # stand-in for the per-rank restoration of images lo..hi-1
        counts = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
        out = gather_restored(local.contiguous(), counts)
        assert out.shape == full.shape, (out.shape, full.shape)
        assert torch.equal(out, full * 2.0 + 1.0), "gathered batch differs"
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""" % ROOT)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_shard_and_gather_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\\n{o}"
        assert f"rank {r} ok" in o


PIPE_WORKER = textwrap.dedent("""
    import os, sys, numpy as np, torch
    import torch.distributed as dist
    sys.path.insert(0, %r)
    from oracle import cases, device_rng as R, models as OM, weights
    from vspbfr_amd.pipeline import shard_range, gather_restored, noise_map_shapes
    from vspbfr_amd import hip_ops as H
    torch.set_grad_enabled(False)
    torch.set_num_threads(3)
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    size, n, seed, step0 = 64, 3, 77, 40           # ragged global batch (2 + 1), images 40..42 of the job
    sd = weights.synth_state_dict("restorenet", weights.load_specs()["restorenet64"], cases.SEED)

    def restore(lo, hi):
        # everything an image consumes is a function of its GLOBAL index lo + b
        B, i0 = hi - lo, step0 + lo
        k = lambda shape, sid, dist_="normal": torch.from_numpy(R.keyed_fill((B,) + tuple(shape[1:]), sid, seed, i0, dist_))
        _, enc_s, dec_s = noise_map_shapes(size, B)
        assert (enc_s, dec_s) == OM.restoration_noise_shapes(size, B)
        lq = k((B, 3, size, size), H.SEG_LQ, "uniform")
        feats = [k((B, 512, 2 ** (j + 2), 2 ** (j + 2)), 200 + j) * 0.5 for j in range(5)]
        pre, z = k((B, 18, 512), H.SEG_XT), k((B, 512), H.SEG_Z)
        en = [k(s, H.SEG_ENC + j) for j, s in enumerate(enc_s)]
        dn = [k(s, H.SEG_DEC + j) for j, s in enumerate(dec_s)]
        return OM.restoration_net(sd, size, lq, feats, pre, [z], en, dn)

    lo, hi = shard_range(n, rank, world)
    counts = [shard_range(n, r, world)[1] - shard_range(n, r, world)[0] for r in range(world)]
    out = gather_restored(restore(lo, hi).contiguous(), counts)
    full = restore(0, n)                           # what one rank computes for the whole batch
    err = float((out - full).abs().max())
    assert out.shape == (n, 3, size, size) and err < 1e-5, err   # CPU conv kernels are not batch-size invariant bit for bit
    assert float((full[0] - full[1]).abs().max()) > 1e-2          # the images really differ
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok", err)
""" % ROOT)


def test_sharded_restoration_equals_single_rank(tmp_path):
    script = tmp_path / "pipe_worker.py"
    script.write_text(PIPE_WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o


REUSE_WORKER = textwrap.dedent("""
    import os, sys, time, torch
    import torch.distributed as dist
    sys.path.insert(0, %r)
    from vspbfr_amd.pipeline import RestoredGather
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    g = RestoredGather()
    static = torch.empty(2, 3, 8, 8)           # ONE buffer reused for every batch, like a captured graph's static output
    pending, got = None, []
    for i in range(5):
        static.fill_(100.0 * i + rank)          # "replay" of batch i overwrites the buffer
        h = g.start(static, stage=True)         # gathered from a copy the gatherer owns
        if rank == 0:
            time.sleep(0.05)                    # a straggling peer: rank 1 runs ahead and overwrites `static` before gather i completes
        if pending is not None:
            got.append(pending.result().clone())
        pending = h
    got.append(pending.result().clone())
    for i, out in enumerate(got):
        assert out.shape == (2 * world, 3, 8, 8)
        for r in range(world):
            assert torch.all(out[2 * r:2 * r + 2] == 100.0 * i + r), (i, r, out[2 * r, 0, 0, 0].item())
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""" % ROOT)


def test_restored_gather_reused_buffer_world2(tmp_path):
    """the all-gather of a buffer the caller overwrites for the next batch (bench.py --graphs hands the graph's static
    output to the asynchronous RestoredGather) must read an owned copy: `start(..., stage=True)`."""
    script = tmp_path / "reuse_worker.py"
    script.write_text(REUSE_WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o


def test_bench_self_launch_world8_ragged():
    """World size 8 (the driver's `bench.py --gpus 8`, BASELINE configs[3]): self-launch + the exchange loop with a ragged last batch over
    eight gloo ranks.  Control path only (--launch-check): the RCCL leg itself has never run on hardware."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "3",
                        "--launch-check"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["launch_check"] == "ok"


def test_bench_self_launch_world4_ragged():
    """World size 4 through the same self-launch: the launch check also runs bench.py's exchange loop -- full batches and a RAGGED
    last one (the last ranks get fewer images) through the preallocated asynchronous RestoredGather, one batch behind."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "2", "--warmup", "1", "--batch", "3",
                        "--launch-check"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 4 and line["launch_check"] == "ok"


def test_bench_self_launch_world2():
    """`python bench.py --gpus 2` with no torchrun environment must start the 2-rank job by itself (as a child process) and
    hand rank 0's JSON line through; --launch-check runs that control path without GPU work."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--launch-check"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["launch_check"] == "ok"


GRAD_WORKER = textwrap.dedent("""
    import os, sys, torch
    import torch.distributed as dist
    sys.path.insert(0, %r)
    from vspbfr_amd.train_step import allreduce_gradients, OverlappedGradientReducer
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.ReLU(), torch.nn.Linear(53, 11), torch.nn.Linear(11, 5))
    x = torch.randn(6, 37, generator=torch.Generator().manual_seed(100 + rank))
    out = net[1](net[0](x))
    out = net[2](out)                       # net[3] takes no part in the loss on any rank: its gradients are None
    out.pow(2).mean().backward()
    if rank == 1:
        net[2].bias.grad = None             # a gradient missing on ONE rank only (find_unused_parameters): it contributes zeros
    mine = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
    nb = allreduce_gradients(list(net.parameters()), bucket_bytes=1024)      # two buckets
    assert nb >= 2, nb
    # reference: gather every rank's gradients and average by hand
    for p, g in zip(net.parameters(), mine):
        g = torch.zeros_like(p) if g is None else g
        parts = [torch.empty_like(g) for _ in range(world)]
        dist.all_gather(parts, g)
        want = sum(parts) / world
        assert p.grad is not None and torch.allclose(p.grad, want, rtol=0, atol=1e-7), (rank, p.shape)
    want = [p.grad.clone() for p in net.parameters()]
    # the hook-driven form (all-reduces started inside backward): the same averaged gradients, bit for bit
    for p in net.parameters():
        p.grad = None
    red = OverlappedGradientReducer(list(net.parameters()), bucket_bytes=1024)
    with red:
        out = net[2](net[1](net[0](x)))
        out.pow(2).mean().backward()
    assert red.launched >= 2, red.launched
    for p, w in zip(net.parameters(), want):
        if p is net[2].bias:                # (the sequential run dropped this gradient on rank 1 by hand)
            continue
        assert p.grad is not None and torch.equal(p.grad, w), (rank, p.shape)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok", nb)
""" % ROOT)


def test_bucketed_gradient_allreduce_world2(tmp_path):
    """train_step.allreduce_gradients (the training step's RCCL gradient exchange; gloo here): flat buckets in reverse
    registration order, averaged, unpacked; parameters without a gradient -- on every rank, or on one rank only -- take part as
    zeros so that all ranks issue the same collectives."""
    script = tmp_path / "grad_worker.py"
    script.write_text(GRAD_WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o


FROZEN_WORKER = textwrap.dedent("""
    import os, sys, torch
    import torch.distributed as dist
    sys.path.insert(0, %r)
    from vspbfr_amd.train_step import OverlappedGradientReducer, RestorationTrainer, requires_grad
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(19, 23), torch.nn.ReLU(), torch.nn.Linear(23, 7))
    requires_grad(net, False)                      # a module left frozen by whoever used it last (an inference pipeline, another trainer)
    red = OverlappedGradientReducer(list(net.parameters()), bucket_bytes=512)
    assert len(red.buckets) >= 2 and sum(len(b) for b in red.buckets) == 4, "every parameter must be bucketed"
    assert not any(p.requires_grad for p in net.parameters()), "the reducer must not thaw the caller's module"
    requires_grad(net, True)
    x = torch.randn(5, 19, generator=torch.Generator().manual_seed(7 + rank))
    net(x).pow(2).mean().backward()
    mine = [p.grad.clone() for p in net.parameters()]
    for p in net.parameters():
        p.grad = None
    with red:
        net(x).pow(2).mean().backward()
    assert red.launched == len(red.buckets)
    for p, g in zip(net.parameters(), mine):
        parts = [torch.empty_like(g) for _ in range(world)]
        dist.all_gather(parts, g)
        assert torch.allclose(p.grad, sum(parts) / world, rtol=0, atol=1e-7), "gradients were not averaged over the ranks"
    # a PARTLY frozen parameter list is refused (ranks that froze different subsets would issue different collectives)
    net[0].weight.requires_grad_(False)
    try:
        OverlappedGradientReducer(list(net.parameters()))
        raise SystemExit("partly frozen parameter list was accepted")
    except ValueError:
        pass
    # the trainer thaws what it is given before it plans its buckets
    G, Ge, D = torch.nn.Linear(4, 4), torch.nn.Linear(4, 4), torch.nn.Linear(4, 1)
    requires_grad(G, False); requires_grad(D, False)
    tr = RestorationTrainer(G, Ge, D)
    assert all(p.requires_grad for p in G.parameters()) and all(p.requires_grad for p in D.parameters())
    assert sum(len(b) for b in tr.g_reducer.buckets) == 2 and sum(len(b) for b in tr.d_reducer.buckets) == 2
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""" % ROOT)


def test_gradient_reducer_on_frozen_modules_world2(tmp_path):
    """a reducer (and a RestorationTrainer) built on modules that are frozen at construction must still exchange every
    gradient once they are thawed; a partly frozen parameter list is refused."""
    script = tmp_path / "frozen_worker.py"
    script.write_text(FROZEN_WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{o}"
        assert f"rank {r} ok" in o
