"""bench.py -- throughput of the restoration hot path on MI355X (contract in the round prompt).

    python bench.py --gpus N --steps K --warmup W

N > 1 without a torchrun environment: bench.py starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a
CHILD process before anything in this process touches the GPU and exits with its code (rank 0's JSON line passes through);
launched by torchrun itself (RANK / WORLD_SIZE set) it is one rank of the job.

One step = one batch through A (e4e encoder) -> B (Code_diffuser DDPM chain) -> C (StyleGAN2 prior decoder, up to
1024^2 as the reference does) -> D (Restoration_net), LQ batch resident in HBM, restored batch left in HBM
(BASELINE.json configs[1]: batch 8 per GPU, 512x512, T = 50, fp32, random-init weights, synthetic inputs).
N > 1: every rank runs its own shard of the global batch (weak scaling: the per-GPU batch is fixed) and the restored images
are all-gathered with RCCL.  Inputs and every noise draw are functions of (seed, GLOBAL image index): the job restores the
same images whatever N is (vsp_keyed_fill_f32).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# the host driver of the GPU boxes supports dmabuf IPC only: without this RCCL's buffer exchange between the ranks of one node fails with
# `hipIpcGetMemHandle: invalid argument` (exported on the boxes already; set here too for a rank started by a bare torchrun)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

PEAK_FP32_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 MFMA (= vector) peak
ALGO_GFLOP_PER_IMAGE = lambda T: 706.9 + 0.455 * T  # noqa: E731  SURVEY.md section 8(d)


def self_launch(n):
    """--gpus N > 1 outside torchrun: run the N-rank job as a child process.  Nothing here may initialise the GPU: replacing or
    forking a process that holds a HIP context is what the GPU boxes forbid (torch.cuda.device_count() does not initialise)."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < n and "--launch-check" not in sys.argv:
        raise SystemExit(f"bench.py: --gpus {n} but only {have} GPU(s) are visible")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


PRESETS = {  # BASELINE.json configs[1..3]
    "c2": dict(batch=8, timesteps=50, sampler="ddpm", conv_dtype="f32"),
    "c3": dict(batch=16, timesteps=50, sampler="ddim", ddim_steps=25, conv_dtype="bf16", act_bf16=True),
    "c4": dict(batch=16, timesteps=50, sampler="ddpm", conv_dtype="f32"),
    # configs[4]: the restoration_train.py iteration, data parallel, 4 images per GPU (batch 32 on 8 GPUs); a different metric
    # (training images/s), reported by train_bench() below -- the default line stays the inference metric of BASELINE.json
    "c5": dict(batch=4, timesteps=4, train=True, steps=16),   # K = 16 = d_reg_every: exactly one R1 pass in the timed region
}


def train_bench(args, world, rank, dev):
    """BASELINE configs[4]: `restoration_train.py` iterations (D step, G step with LPIPS-VGG x 0.5 + ArcFace ID x 0.1, Adam, EMA; the
    R1 regulariser on its every-16th schedule), frozen front through the inference kernels, one process per GPU, gradients
    all-reduced over RCCL from inside backward (train_step.OverlappedGradientReducer).  Random-init networks, keyed synthetic batch.
    Same timing contract as the inference line: W warm-up iterations, K timed ones between barrier + synchronize, max over ranks."""
    import copy
    import torch.distributed as dist
    from vspbfr_amd import hip_ops
    from vspbfr_amd.discriminator import Discriminator
    from vspbfr_amd.id_loss import IDLoss
    from vspbfr_amd.lpips import PerceptualLoss
    from vspbfr_amd.train_step import RestorationTrainer
    B = args.batch
    pipe = build_pipeline(dev, args.timesteps, False)          # T = 4: load_ddpm's default in restoration_train.py
    G = pipe.generator
    torch.manual_seed(1)                                        # the same initial weights on every rank
    D = Discriminator(512).to(dev)
    tr = RestorationTrainer(G, copy.deepcopy(G), D, psp_embedding=pipe.psp, diffusion=pipe.diffusion, mixing=0.9,
                            percept_loss=PerceptualLoss().to(dev), percept_weight=0.5, id_loss=IDLoss(None, device=dev), id_weight=0.1)
    low, real = hip_ops.keyed_fill([(B, 3, 512, 512), (B, 3, 512, 512)], [hip_ops.SEG_LQ, 60], args.seed, rank * B,
                                   dist="uniform", device=dev)
    G.train()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    it = [0]

    def run(n):
        for _ in range(n):
            it[0] += 1
            tr.step(it[0], low, real)                           # iterations 1 .. : the R1 pass falls on every 16th
    run(args.warmup)
    prof = hip_ops.ConvProfiler()
    hip_ops.PROFILER = prof
    run(1)
    hip_ops.PROFILER = None
    sync()
    fl, conv_ms, n_launch = prof.summary()
    ex_fl = prof.executed_flops()      # what the matrix pipe ran: Winograd launches at 16/36 (F(2x2)) or 36/144 (F(4x4)) of their algorithmic count
    # The R1 pass (a double backward through D) runs on every d_reg_every-th iteration: the timed region starts right after a
    # multiple of it, so that K timed iterations contain exactly floor(K / d_reg_every) of them -- one in sixteen with the default
    # K = 16 of this preset, as in the schedule being priced; the count is stated in the line.
    while it[0] % tr.d_reg_every:
        run(1)
    sync()
    first = it[0] + 1
    n_r1 = sum(1 for i in range(first, first + args.steps) if i % tr.d_reg_every == 0)
    t0 = time.perf_counter()
    run(args.steps)
    sync()
    dt = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    dt = float(dt)
    if rank == 0:
        ms = dt / args.steps * 1e3
        print(json.dumps({
            "metric": "restoration_train.py images/sec (512x512, RestoreNet + Discriminator fwd/bwd + LPIPS + id_loss + Adam + EMA)",
            "value": round(world * B / ms * 1e3, 3), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: restoration_train.py iteration, %d images per GPU, 512x512, frozen e4e + "
                                   "Code_diffuser(T=4) + StyleGAN2 prior front, random-init weights, RCCL gradient all-reduce "
                                   "from inside backward (64 MB buckets)" % B, "global_batch": world * B},
            "r1": {"d_reg_every": tr.d_reg_every, "r1_iterations_in_timed_region": n_r1,
                   "note": "the timed region starts right after a multiple of d_reg_every; K a multiple of it prices the schedule exactly"},
            "roofline": {"bound": "mfma", "achieved": round(ex_fl / conv_ms / 1e9, 1), "peak": 157.3, "unit": "TFLOP/s",
                         "frac": round(ex_fl / conv_ms / 1e9 / 157.3, 3), "traffic": None,
                         "algorithmic_tflops": round(fl / conv_ms / 1e9, 1), "algorithmic_frac": round(fl / conv_ms / 1e9 / 157.3, 3),
                         "kernel": "conv family of one iteration (forward, data gradient, weight gradient, loss networks); achieved / frac = FLOPs "
                                   "the matrix pipe EXECUTED / kernel time (Winograd launches run 16/36 or 36/144 of their direct-form count); "
                                   "algorithmic_* = direct-form FLOPs / time (an effective rate)",
                         "launches": n_launch, "kernel_ms": round(conv_ms, 1),
                         "measured": "HIP events per launch on the launch stream over one untimed iteration"},
            "cpu_baseline": None}))
    if world > 1:
        dist.destroy_process_group()
    return 0


def launch_check(args, world, rank):
    """--launch-check: the job's control path without a GPU (tests/test_distributed_cpu.py): ranks rendezvous over gloo, every
    rank takes its shard of a stand-in global batch whose values encode the GLOBAL image index, the shards are all-gathered
    by the path's own gather_restored and rank 0 prints the JSON line (value null: nothing was measured)."""
    import torch.distributed as dist
    from vspbfr_amd.pipeline import gather_restored, shard_range
    if world > 1:
        dist.init_process_group("gloo")
    B = args.batch
    lo, hi = shard_range(world * B, rank, world)
    assert (lo, hi) == (rank * B, (rank + 1) * B)
    local = torch.arange(lo, hi, dtype=torch.float32).view(-1, 1, 1, 1).expand(-1, 3, 8, 8).contiguous()
    full = gather_restored(local) if world > 1 else local
    ok = torch.equal(full[:, 0, 0, 0], torch.arange(world * B, dtype=torch.float32))
    # the loop bench.py runs: `steps` full batches and a RAGGED last one (world * B - 3 images: the last rank(s) get fewer), each
    # exchanged by the preallocated asynchronous gather one batch behind the computation
    from vspbfr_amd.pipeline import RestoredGather
    gat, pending, got = RestoredGather(), None, []
    sizes = [world * B] * args.steps + [max(world * B - 3, 1)]
    base = 0
    for n_img in sizes:
        lo, hi = shard_range(n_img, rank, world)
        counts = [shard_range(n_img, r, world)[1] - shard_range(n_img, r, world)[0] for r in range(world)]
        mine = (base + torch.arange(lo, hi, dtype=torch.float32)).view(-1, 1, 1, 1).expand(-1, 3, 8, 8).contiguous()
        h = gat.start(mine, counts if len(set(counts)) > 1 else None)
        if pending is not None:
            got.append(pending[0].result().clone())
        pending = (h, n_img)
        base += n_img
    got.append(pending[0].result().clone())
    seen = torch.cat([g[:, 0, 0, 0] for g in got])
    ok = ok and torch.equal(seen, torch.arange(sum(sizes), dtype=torch.float32)) and len(gat._out) <= 4
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "restored 512x512 faces/sec", "value": None, "unit": "img/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "launch_check": "ok" if ok else "gathered batch out of order"}), flush=True)
    return 0 if ok else 1


def build_pipeline(dev, T, with_sample, noise_seed=None):
    from vspbfr_amd.diffusion import Code_diffuser, My_DDPM
    from vspbfr_amd.e4e import E4e_embedding, Encoder4Editing, Generator
    from vspbfr_amd.pipeline import RestorationPipeline
    from vspbfr_amd.restorenet import Restoration_net
    from argparse import Namespace
    torch.manual_seed(0)
    gen = Restoration_net(512, 512, 8, channel_multiplier=2)
    net = Code_diffuser(timesteps=T)
    enc = Encoder4Editing(50, "ir_se", Namespace(input_channel=3, stylegan_size=1024))
    dec = Generator(1024, 512, 8, channel_multiplier=2)
    sd = {"encoder." + k: v for k, v in enc.state_dict().items()}
    sd.update({"decoder." + k: v for k, v in dec.state_dict().items()})
    ckpt = {"state_dict": sd, "latent_avg": 0.1 * torch.randn(18, 512),
            "opts": {"encoder_type": "Encoder4Editing", "stylegan_size": 1024, "start_from_latent_avg": True}}
    psp = E4e_embedding(ckpt, out_size=512, size=1024, device=dev, use_generator=True)
    ddpm = My_DDPM(denoise=net.to(dev).eval(), timesteps=T).to(dev)  # default betas (1e-4, 2e-2), SURVEY 8a row 8
    return RestorationPipeline(gen.to(dev).eval(), psp, ddpm, mixing=0.0, with_sample=with_sample, noise_seed=noise_seed)


def cpu_baseline(T, threads, sample_steps=None):
    """The CPU restatement of the same step (oracle/, kind = "port": the Python reference cannot travel to the GPU box),
    timed on the host cores on a BOUNDED sample: ONE image through A, B, C, D in full (round 4: the whole T-step chain runs --
    a step of the 18 x 512 token network costs ~7 ms on the host, so nothing is extrapolated any more; `sample_steps` < T
    restores the sampled form: every step costs the same, 4 TACC blocks)."""
    sample_steps = T if sample_steps is None else min(sample_steps, T)
    from oracle import pipeline as OP
    torch.set_num_threads(threads)
    ck = OP.synth_checkpoints()
    inp = OP.draw_inputs("bench_cpu", 1)
    st = {}
    OP.restore(ck, inp, timesteps=sample_steps, linear_start=1e-4, linear_end=2e-2, timings=st)
    per_step = st["diffuser"] / sample_steps
    total = st["encoder"] + st["prior_decoder"] + st["restorenet"] + per_step * T
    measured = sum(st.values())
    ref = None
    ref_path = os.path.join(ROOT, "profiles", "r02_reference_cpu_timing.json")
    if os.path.exists(ref_path):  # the reference ITSELF, timed in the build container (tools/time_reference_cpu.py; it cannot travel here)
        with open(ref_path) as f:
            r = json.load(f)
        ref = {"what": r["what"], "threads": r["threads"],
               "B4_T10_img_per_s": r.get("c1", {}).get("img_per_s"), "B1_T50_img_per_s": r.get("b1t50", {}).get("img_per_s")}
    ref4_path = os.path.join(ROOT, "profiles", "r04_reference_cpu_timing.json")
    if os.path.exists(ref4_path):  # round 4: the reference at the configuration the metric is quoted on (BASELINE configs[1]: B = 8, T = 50)
        with open(ref4_path) as f:
            r4 = json.load(f)
        ref = dict(ref or {"what": r4["what"], "threads": r4["threads"]})
        ref["C2_B8_T50_img_per_s"] = r4.get("c2", {}).get("img_per_s")
        ref["C2_B8_T50_s_per_batch"] = r4.get("c2", {}).get("s_per_batch")
    return {"value": round(1.0 / total, 5), "unit": "img/s", "cores": threads, "kind": "port",
            "sample": (f"1 image 512x512 through stages A, B (all {T} chain steps), C (to 1024^2), D in full: {total:.1f} s/image measured"
                       if sample_steps == T else
                       f"1 image 512x512: stages A, C (to 1024^2), D in full + {sample_steps} of {T} DDPM steps ({measured:.1f} s "
                       f"measured), chain extrapolated linearly to T={T} -> {total:.1f} s/image"),
            "reference_in_build_container": ref,
            "stage_seconds": {"encoder": round(st["encoder"], 2), "diffuser_per_step": round(per_step, 3),
                              "prior_decoder": round(st["prior_decoder"], 2), "restorenet": round(st["restorenet"], 2)}}


def conv_traffic(B, args):
    """HBM bytes per conv launch from the committed PMC passes (FETCH_SIZE x2 per the gfx950 calibration, WRITE_SIZE x1, separate
    passes: tools/pmc_bench.sh -> tools/pmc_traffic_summary.py).  PMC collection cannot run inside the timed process, so the
    figure is only reported for the configurations it was measured on: C2 (batch 8, T = 50 DDPM, fp32) and C3 (batch 16, DDIM 25,
    bf16 kernels + bf16 activations)."""
    if args.no_sample:
        return None
    tag = None
    if args.conv_dtype == "f32" and B == 8 and args.timesteps == 50 and args.sampler == "ddpm":
        tag = "c2_f32"
    elif args.conv_dtype == "bf16" and args.act_bf16 and B == 16 and args.sampler == "ddim" and args.ddim_steps == 25:
        tag = "c3_bf16act"
    elif args.conv_dtype == "bf16" and not args.act_bf16 and B == 8 and args.timesteps == 50 and args.sampler == "ddpm":
        tag = "b8_bf16"
    for name in ({"c2_f32": ["r06_conv_traffic_pmc_c2_f32.json", "r05_conv_traffic_pmc_c2_f32.json", "r04_conv_traffic_pmc_c2_f32.json", "r03_conv_traffic_pmc_c2_f32.json", "r02_conv_traffic_pmc_c2_f32.json", "r01_conv_traffic_pmc.json"], "c3_bf16act": ["r06_conv_traffic_pmc_c3_bf16act.json", "r05_conv_traffic_pmc_c3_bf16act.json", "r04_conv_traffic_pmc_c3_bf16act.json", "r02_conv_traffic_pmc_c3_bf16act.json"],
                  "b8_bf16": ["r01_conv_traffic_pmc_bf16.json"]}.get(tag, [])):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                return round(json.load(f)["conv_hbm_bytes_per_launch"])
    return None


def measure(args, world, rank, dev, pipe=None):
    """One configuration through the timing contract (W warm-up steps, one serial step with per-launch events for the roofline figures, K timed
    steps between barrier + synchronize, max over ranks).  Returns (JSON line as a dict on rank 0 else None, the pipeline for re-use)."""
    import torch.distributed as dist
    from vspbfr_amd import hip_ops
    hip_ops.BF16_CONV = {"f32": False, "bf16": True, "bf16x3": "x3"}[args.conv_dtype]
    if args.act_bf16 and args.conv_dtype != "bf16":
        raise SystemExit("--act-bf16 needs --conv-dtype bf16")
    if pipe is None:
        pipe = build_pipeline(dev, args.timesteps, not args.no_sample, None if args.torch_rng else args.seed)
    pipe.act_bf16 = bool(args.act_bf16)
    pipe.encoder_fp32, pipe.encoder_x3 = args.encoder == "f32", args.encoder == "x3"
    ddpm_module = pipe.diffusion
    if args.sampler == "ddim":
        from vspbfr_amd.ddim import DDIMSampler
        ddpm, S = pipe.diffusion, args.ddim_steps
        sampler = DDIMSampler(ddpm, device=dev)

        class _DDIM(torch.nn.Module):  # same call shape as My_DDPM.forward for the pipeline
            def forward(self, x=None, condi_in=None, training=False, x_T=None):
                return sampler.sample(S=S, batch_size=condi_in.shape[0], shape=18 * 512, conditioning=condi_in, eta=0.0,
                                      verbose=False, x_T=x_T)[0]
        pipe.diffusion = _DDIM()
    B = args.batch
    # this rank's shard of the global LQ batch: images [rank*B, (rank+1)*B), uniform(-1, 1) keyed by the global image index
    lq = hip_ops.keyed_fill([(B, 3, 512, 512)], [hip_ops.SEG_LQ], args.seed, rank * B, dist="uniform", device=dev)[0]
    step_no = [0]
    from vspbfr_amd.pipeline import RestoredGather
    gatherer = RestoredGather()   # preallocated, asynchronous: the exchange of batch i runs under C + D of batch i+1

    def batches(n):
        """n steps: the same LQ shard with fresh noise -- step k restores global images [k*W*B, (k+1)*W*B)."""
        for _ in range(n):
            k = step_no[0]
            step_no[0] += 1
            yield lq, (k * world + rank) * B

    def run(n):
        """n steps = n batches through A+B+C+D.  --overlap (default): RestorationPipeline.run_batches, i.e. stages A+B of
        batch i+1 are enqueued on a second HIP stream before stages C+D of batch i (the first batch's A+B is not hidden);
        every batch is complete when the closing synchronize returns."""
        res, pending = None, None

        def exchange(restored, reused=False):
            """start this batch's all-gather behind its kernels, then collect the PREVIOUS batch's (which ran under this batch's C + D);
            reused = `restored` is a buffer the next batch overwrites (a captured graph's static output): gathered from an owned copy"""
            nonlocal res, pending
            if world == 1:
                res = restored
                return
            h = gatherer.start(restored, stage=reused)
            if pending is not None:
                res = pending.result()
            pending = h
        if args.no_overlap:
            for x, i0 in batches(n):
                exchange(pipe(x, image_index0=i0)["restored"])
        elif args.graphs:
            for o in pipe.run_batches_graphed(batches(n)):
                exchange(o["restored"], reused=True)
        else:
            for o in pipe.run_batches(batches(n)):
                exchange(o["restored"])
        if pending is not None:
            res = pending.result()
        return res

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        if args.graphs:
            pipe.capture_graphs(lq)
        run(args.warmup)
        iso = None
        if not args.no_overlap:
            # one extra UNTIMED serial step with per-launch events: the conv kernels' durations with nothing else on the GPU
            # (in the timed, overlapped region a conv's event interval also contains the side stream's kernels)
            pipe(lq)  # the serial call path allocates from the main stream's pool: warm it before measuring
            torch.cuda.synchronize()
            iso = hip_ops.ConvProfiler()
            hip_ops.PROFILER = iso
            pipe(lq)
            hip_ops.PROFILER = None
            torch.cuda.synchronize()
        # Per-launch events inside the timed region only for the serial loop (--no-overlap), where they ARE the kernel durations; the
        # default overlapped loop takes every roofline figure from the `iso` step above and times K clean steps.
        prof = hip_ops.ConvProfiler() if iso is None else None
        hip_ops.PROFILER = prof
        sync()
        t0 = time.perf_counter()
        res = run(args.steps)
        sync()
        dt = time.perf_counter() - t0
        hip_ops.PROFILER = None
    assert torch.isfinite(res).all(), "non-finite output"
    tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if prof is not None:
        conv_flops, conv_ms, conv_launches = prof.summary()
        conv_exec = prof.executed_flops()
    else:   # the serial measurement step, scaled to the K timed steps (the line's per-step fields divide by K again)
        conv_flops, conv_ms, conv_launches = (v * args.steps for v in iso.summary())
        conv_exec = iso.executed_flops() * args.steps

    pipe.diffusion = ddpm_module   # (a DDIM configuration wrapped it)
    pipe.encoder_fp32 = pipe.encoder_x3 = False
    line = None
    if rank == 0:
        imgs = world * B * args.steps
        achieved = conv_flops / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
        # bf16 configuration: the dense bf16 MFMA peak (MI355X_MICROARCH.md); the conv family then mixes bf16 (stride-1 3x3)
        # and fp32 (stride-2, transposed, small-map) launches, all priced against the bf16 peak
        PEAK = {"f32": PEAK_FP32_TFLOPS, "bf16": 2500.0, "bf16x3": 2500.0 / 3}[args.conv_dtype]  # bf16x3: three MFMAs per product
        KERNEL_NOTE = ("conv family: conv_pipe_kernel (direct, double-buffered pipeline: stride-2 / transposed) + "
                       "conv_igemm_kernel / conv_smallmap_kernel (direct, small maps and 1x1) + conv_wino_ro_kernel / conv_wino_rod_kernel / "
                       "conv_wino_rs_kernel / conv_wino_kernel (Winograd F(2x2,3x3): row-owner, dilation groups, register-resident U) + wino4_input_kernel / wino4_gemm_kernel (Winograd F(4x4,3x3), deep layers) + conv_wino4f_kernel / conv_wino4f_groups_kernel (Winograd F(4x4,3x3) fused in registers: shallow wide layers, dilation groups of 128 - 512 channels); "
                       "achieved / frac = FLOPs the matrix pipe EXECUTED / kernel time (F(2x2) launches run 16/36, F(4x4) launches 36/144 of their "
                       "direct-form count); algorithmic_tflops / algorithmic_frac = direct-form FLOPs / time, an effective rate on the Winograd layers") if args.conv_dtype == "f32" else (
            "conv family: conv_bf16_kernel (bf16 MFMA 32x32x16, fp32 accumulate: stride-1 / stride-2 / transposed 3x3 layers; modulated layers on per-image weights bf16(W * style), copy-only staging) + conv_bf16_dg_kernel (64 -> 4 x 16 dilation groups) + conv_bf16_rv_kernel "
            "(row-vector K: plain stride-1 layers with <= 256 channels on maps >= 128^2, bf16 activations) + the fp32 "
            "conv_igemm_kernel on small maps and 1x1 layers; achieved = algorithmic FLOPs / time against the dense bf16 MFMA peak; with "
            "fp32 activations in HBM the 512^2 / 256^2 layers are fabric-bound (DESIGN 9)")
        line = {
            "metric": "restored 512x512 faces/sec", "value": round(imgs / dt, 3), "unit": "img/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.conv_dtype, "data": "synthetic",
            "config": {"workload": f"restoration_test.py hot path A+B+C+D, batch {B}/GPU, 512x512, {args.timesteps}-step DDPM "
                                   f"CodeDiffuser + StyleGAN2 prior{'' if not args.no_sample else ' (no 1024^2 tail)'} + RestoreNet "
                                   "forward, " + {"f32": "fp32", "bf16": "bf16-MFMA convolutions (fp32 accumulate), " + (
                                       "bf16 activations in HBM for stages C + D" if args.act_bf16 else "fp32 activations") + ", rest fp32",
                                                 "bf16x3": "split-precision bf16-MFMA convolutions (hi + lo operand pairs, 3 MFMAs per product, fp32 "
                                                           "accumulate; transposed and small-map layers fp32), rest fp32"}[args.conv_dtype] + ", random-init weights",
                       "batch_per_gpu": B, "timesteps": args.timesteps, "with_style_sample": not args.no_sample,
                       "sampler": args.sampler if args.sampler == "ddpm" else f"ddim S={args.ddim_steps}",
                       "overlap": "none" if args.no_overlap else "A+B of batch i+1 on a second HIP stream under C+D of batch i",
                       "launch": "two captured HIP graphs (A+B, C+D) replayed per batch" if args.graphs else "eager (one host launch per kernel)",
                       "rng": "torch device RNG, one randn per consumer" if args.torch_rng else
                              "keyed Philox draws by (seed, global image index): 2 launches per batch; the DRAWS are world-size invariant (mixing = 0: no per-batch style-mixing coin), the kernels' summation orders follow the per-rank batch size",
                       "sharding": f"dp{world}: batch split, weights replicated, all-gather of restored images" if world > 1 else "single GPU"},
            # achieved / frac = what the matrix pipe EXECUTED per second (<= peak by construction: a Winograd F(2x2,3x3) launch runs 16/36, an
            # F(4x4,3x3) launch 36/144 of its direct-form count); algorithmic_* = direct-form FLOPs / time, an effective rate that may exceed 1
            "roofline": {"bound": "mfma", "achieved": round(conv_exec / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0, 2), "peak": PEAK, "unit": "TFLOP/s",
                         "frac": round((conv_exec / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0) / PEAK, 4),
                         "algorithmic_tflops": round(achieved, 2), "algorithmic_frac": round(achieved / PEAK, 4),
                         "traffic": conv_traffic(B, args),
                         "kernel": KERNEL_NOTE, "launches_per_step": conv_launches // max(args.steps, 1),
                         "algorithmic_gflop_per_step": round(conv_flops / max(args.steps, 1) / 1e9, 1),
                         "kernel_ms_per_step": round(conv_ms / max(args.steps, 1), 2),
                         "pipeline_frac_of_fp32_peak": round(imgs / dt * ALGO_GFLOP_PER_IMAGE(args.timesteps) / 1e3 / (PEAK * world), 4)},
        }
        if args.conv_dtype == "bf16":
            # BASELINE configs[2] / SURVEY 8d: with the bf16 matrix pipe (2.5 PFLOP/s) the conv family is bound by HBM, not by MFMA.
            # achieved = ALGORITHMIC bytes (every operand of a launch crosses HBM once at its element size: input, output, weights,
            # noise, residuals) / kernel time, against the 8 TB/s HBM3E peak; the MFMA figure stays in the line as `mfma`.
            src = iso if iso is not None else prof
            fl, ms, n = src.summary()
            by = src.algorithmic_bytes()
            if iso is None:  # the timed region holds K steps
                fl, ms, n, by = fl / args.steps, ms / args.steps, n // args.steps, by / args.steps
            rl = line["roofline"]
            rl["mfma"] = {"achieved_tflops": round(fl / (ms * 1e-3) / 1e12, 2), "peak": PEAK, "frac": round(fl / (ms * 1e-3) / 1e12 / PEAK, 4)}
            rl.update({"bound": "hbm", "achieved": round(by / (ms * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                       "frac": round(by / (ms * 1e-3) / 1e9 / 8000.0, 4), "algorithmic_gb_per_step": round(by / 1e9, 2),
                       "launches_per_step": n, "kernel_ms_per_step": round(ms, 2),
                       "activations": "bf16 in HBM (2 B per element)" if args.act_bf16 else "fp32 in HBM (converted while staged)",
                       "measured": "HIP events per launch on the launch stream over one serial step inside bench.py" if iso is not None
                       else "HIP events per launch inside the timed region (no second stream: --no-overlap)"})
            rl.pop("algorithmic_gflop_per_step", None)
        elif iso is not None:
            # The roofline figures of the kernel are the per-launch HIP-event durations of ONE serial step taken right before the timed
            # region: in the timed region two streams run at once (an event interval there would also contain the other stream's
            # kernels), so it carries no per-launch events.  rocprofv3 serialises dispatches: its per-kernel durations (profiles/) agree
            # with these.
            fl, ms, n = iso.summary()
            ex = iso.executed_flops()
            rl = line["roofline"]
            rl["achieved"], rl["frac"] = round(ex / (ms * 1e-3) / 1e12, 2), round(ex / (ms * 1e-3) / 1e12 / PEAK, 4)
            rl["algorithmic_tflops"], rl["algorithmic_frac"] = round(fl / (ms * 1e-3) / 1e12, 2), round(fl / (ms * 1e-3) / 1e12 / PEAK, 4)
            rl["launches_per_step"], rl["algorithmic_gflop_per_step"], rl["kernel_ms_per_step"] = n, round(fl / 1e9, 1), round(ms, 2)
            rl["executed_gflop_per_step"] = round(ex / 1e9, 1)
            rl["by_kernel_family"] = {k: {"algorithmic_tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 1), "ms": round(v[1], 2), "launches": v[2]}
                                      for k, v in sorted(iso.by_kind().items()) if v[1] > 0}
            rl["measured"] = ("HIP events per launch on the launch stream over one serial step inside bench.py, right before the "
                              "timed region (no second stream in flight; the timed region itself carries no per-launch events)")
    hip_ops.BF16_CONV = False
    return line, pipe


def extra_configs(args, world, rank, dev, pipe):
    import copy
    out = {}
    for name, key in (("c3", "c3"), ("c4", "c4_share")):
        a = copy.copy(args)
        for k, v in PRESETS[name].items():
            setattr(a, k, v)
        a.preset = name
        try:
            ln, _ = measure(a, world, rank, dev, pipe)
            out[key] = {k: ln[k] for k in ("value", "unit", "ms_per_step", "steps", "warmup", "dtype", "config", "roofline")}
        except Exception as e:   # the headline line must survive whatever an extra configuration does
            out[key] = {"error": f"{type(e).__name__}: {e}"[:400]}
            torch.cuda.synchronize()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)   # with the batch overlap the first batch's A+B is not hidden: K = 3 reads ~3 % low
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step (BASELINE configs[1]: 8)")
    ap.add_argument("--timesteps", type=int, default=50)
    ap.add_argument("--no-sample", action="store_true", help="skip the 1024^2 tail of the prior (not the headline config)")
    ap.add_argument("--sampler", choices=["ddpm", "ddim"], default="ddpm", help="ddim = BASELINE configs[2]'s sampler (fp32 here)")
    ap.add_argument("--ddim-steps", type=int, default=25)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="plain per-batch loop (no A+B / C+D stream overlap across batches)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--graphs", action="store_true",
                    help="replay stages A+B and C+D as two captured HIP graphs (same kernels, same two-stream overlap, no per-launch host work)")
    ap.add_argument("--conv-dtype", choices=["f32", "bf16", "bf16x3"], default="f32",
                    help="bf16 = BASELINE configs[2]'s kernels: eligible convolutions on vsp_conv2d_bf16 (bf16 MFMA, fp32 accumulate, "
                         "fp32 activations in HBM); not the parity configuration.  bf16x3 = the same kernels with hi + lo bf16 "
                         "operand pairs and three MFMAs per product (vsp_conv2d_bf16x3): fp32-grade results on the bf16 pipe")
    ap.add_argument("--act-bf16", action="store_true",
                    help="with --conv-dtype bf16: bf16 ACTIVATIONS in HBM between the kernels of stages C + D (every map of 32^2 and "
                         "larger; vsp_conv2d_bf16 io_bf16, vsp_upfirdn2d_bf16, vsp_pointwise_bf16) -- BASELINE configs[2] as specified")
    ap.add_argument("--preset", choices=sorted(PRESETS), default=None,
                    help="BASELINE.json configuration: c2 = batch 8, 50-step DDPM, fp32 (the default); c3 = batch 16, DDIM 25, bf16 "
                         "kernels; c4 = batch 16 per GPU (128 on 8 GPUs), 50-step DDPM, fp32")
    ap.add_argument("--seed", type=int, default=123, help="seed of the keyed input / noise draws")
    ap.add_argument("--torch-rng", action="store_true", help="draw noise from torch's device RNG stream (one randn per consumer, "
                                                              "as the reference does) instead of the keyed single-launch draws")
    ap.add_argument("--encoder", choices=["same", "f32", "x3"], default="same",
                    help="bf16 configurations: the kernels of stage A (e4e encoder) -- same as stages C + D (default), fp32, or the split-precision "
                         "bf16 kernels (fp32-grade codes for the sampler chain, which amplifies the encoder's rounding)")
    ap.add_argument("--no-extra", action="store_true", help="headline configuration only: skip the extra_configs legs (configs[2], the one-GPU "
                                                             "share of configs[3]) that a default one-GPU run appends to its JSON line")
    ap.add_argument("--launch-check", action="store_true", help="no GPU work: self-launch, rendezvous (gloo), shard + all-gather "
                                                                 "of a stand-in batch, JSON line with value null")
    args = ap.parse_args()
    if args.preset:
        given = {a.split("=")[0] for a in sys.argv[1:] if a.startswith("--")}
        for k, v in PRESETS[args.preset].items():
            if k == "steps" and "--steps" in given:   # (a preset's default K does not override an explicit one)
                continue
            setattr(args, k, v)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.launch_check:
        raise SystemExit(launch_check(args, world, rank))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    if getattr(args, "train", False):
        raise SystemExit(train_bench(args, world, rank, dev))
    line, pipe = measure(args, world, rank, dev)
    if rank == 0:
        if world == 1 and not args.no_extra and args.preset in (None, "c2") and args.conv_dtype == "f32" and args.sampler == "ddpm" and not (
                args.no_overlap or args.graphs or args.no_sample or args.torch_rng):
            # BASELINE configs[2] and the one-GPU share of configs[3], measured the same way in the same process AFTER the headline's timed
            # region (the line's value / config / dtype stay configs[1]); same random-init networks, their own warm-up
            line["extra_configs"] = extra_configs(args, world, rank, dev, pipe)
        if world == 1 and not args.no_cpu_baseline:
            threads = args.cpu_threads or min(os.cpu_count() or 1, 16)
            line["cpu_baseline"] = cpu_baseline(args.timesteps, threads)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
