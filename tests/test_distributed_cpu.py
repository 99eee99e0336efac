"""Multi-process check of the data-parallel plumbing (gloo, world_size 2, CPU): the batch split covers every image once,
ragged splits work, and the all-gather reassembles the restored batch in rank order on every rank -- the same code path
bench.py / the CLI use with RCCL on the GPUs."""
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
