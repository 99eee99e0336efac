"""The restoration hot path A -> B -> C -> D (reference restoration_test.py:125-131) as one object, plus the
data-parallel sharding used across the GPUs of a node.

    low_latent     = psp_embedding.get_w_plus(low_imgs)                      # A  e4e encoder
    pre_dic_latent = diffusion(x=low_latent, condi_in=low_latent)            # B  Code_diffuser DDPM chain
    sample, feats  = psp_embedding.get_stylegan_feats(pre_dic_latent)        # C  StyleGAN2 prior decoder
    restored       = generator(low_imgs, feats, pre_dic_latent, noise)       # D  Restoration_net

Every image is independent end to end (eval-mode BatchNorm, per-sample modulated convs), so multi-GPU is a
contiguous split of the batch with replicated weights and ONE all-gather of the restored images (RCCL over xGMI via
torch.distributed backend "nccl"); there is no collective inside the path.
"""
import torch

from ._lib import tune_env
from .diffusion import Code_diffuser, My_DDPM
from .e4e import E4e_embedding
from .restorenet import Restoration_net, mixing_noise


def _side_stream():
    """The second HIP stream of the batch loop (stages A + B of the next batch), at HIGH priority: its ~700 short launches form a
    latency chain, and dispatched ahead of the main stream's big convolutions whenever a slot frees they finish early instead of
    trailing into the next batch (round 5, same box, --steps 20: 190.0 / 190.6 img/s against 188.8 / 189.2 at the default
    priority).  VSP_SIDE_PRIORITY = 0 restores the default."""
    import os
    pr = int(tune_env("VSP_SIDE_PRIORITY", "-1"))
    return torch.cuda.Stream(priority=pr) if pr else torch.cuda.Stream()


def _tensors(obj):
    """every tensor inside nested tuples / lists / dicts"""
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors(v)
    elif isinstance(obj, (tuple, list)):
        for v in obj:
            yield from _tensors(v)


def shard_range(n, rank, world):
    """Contiguous split of n items: rank r owns [lo, hi); sizes differ by at most one (ragged batches allowed)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def load_ddpm(ddpm_ckpt, device="cuda", timesteps=4, linear_start=0.1, linear_end=0.99):
    """reference restoration_test.py:31-40 (checkpoint key "att_mapper"); `ddpm_ckpt` may be a path or a state dict."""
    sd = ddpm_ckpt if isinstance(ddpm_ckpt, dict) else torch.load(ddpm_ckpt, map_location="cpu")
    if "att_mapper" in sd and not any(k.startswith("att_mapper.") for k in sd):
        sd = sd["att_mapper"]
    net = Code_diffuser(timesteps=timesteps).to(device)
    net.load_state_dict(sd)
    net.eval()
    return My_DDPM(denoise=net, linear_start=linear_start, linear_end=linear_end, timesteps=timesteps).to(device)


def noise_map_shapes(size, batch, gen_size=None):
    """Shapes of the NoiseInjection draws in call order: (prior decoder [4, then (r, r) for r = 8..gen_size] -- reference
    e4e/models/stylegan2/model.py:526-540; Restoration_net encoder [SMART at r, down-conv at r/2 for r = size..8] and decoder
    [4, then (r, r) for r = 8..size] -- models/RestoreNet.py:922-927, 1022-1037)."""
    import math
    ls = int(math.log2(size))
    enc = []
    for i in range(ls, 2, -1):
        enc += [(batch, 1, 2 ** i, 2 ** i), (batch, 1, 2 ** (i - 1), 2 ** (i - 1))]
    pyramid = lambda top: [(batch, 1, 4, 4)] + [(batch, 1, 2 ** i, 2 ** i) for i in range(3, top + 1) for _ in (0, 1)]  # noqa: E731
    return pyramid(int(math.log2(gen_size or size))), enc, pyramid(ls)


class RestorationPipeline:
    def __init__(self, generator: Restoration_net, psp_embedding: E4e_embedding, diffusion: My_DDPM, mixing=0.5,
                 with_sample=True, noise_seed=None):
        """noise_seed=None: every random draw comes from torch's device RNG stream, one `randn` per consumer, as in the
        reference.  noise_seed=int: KEYED draws (hip_ops.keyed_fill): x_T, z and all 46 noise maps of a batch are functions of
        (noise_seed, global image index, tensor id) drawn by two launches (one per stage pair) -- with mixing = 0 (bench, CLI
        default of this repository's tests) the result for an image does not depend on the batch it travels in or on the rank that
        computes it -- up to the last bits of the kernels' summation orders, which are chosen per launch SHAPE: the few-row GEMM
        form (M <= 16 rows: every style modulation at batch <= 16) sums K in another order than the MFMA tile form (M > 16), and the
        conv tile / Winograd variant comes from a per-shape table keyed by the batch size, so an image restored inside a batch of 32
        differs from the same image inside a batch of 4 by fp32 rounding noise (1e-6 relative per layer; the draws themselves are
        bit-identical).  Ranks of one job run the same batch size, so the all-gathered result does not depend on the world size at a fixed
        per-rank batch.  With mixing > 0 the invariance does NOT hold at all: the reference flips ONE style-mixing coin and draws
        ONE inject index per BATCH (restoration_test.py:77-82), which is kept here and keyed by the batch's first global image index,
        so an image's result then depends on which batch it is in; graph capture refuses mixing > 0 for the same reason (the coin
        would be frozen into the graph)."""
        self.generator, self.psp, self.diffusion = generator.eval(), psp_embedding.eval(), diffusion.eval()
        self.mixing, self.with_sample, self.noise_seed = mixing, with_sample, noise_seed
        self._index_tensor = None  # device int64 base index (graph replays)
        # bf16 ACTIVATIONS in HBM for stages C + D (BASELINE configs[2]; only with hip_ops.BF16_CONV = True): every map of 32^2
        # and larger travels between the kernels as bf16, the 3-channel images, the latents, the noise maps and stage A + B stay fp32
        self.act_bf16 = False
        # option for the split-precision configuration: keep the encoder on the fp32 kernels.  Off: measured on the pinned case the
        # free-running result is the same either way (codes 2.5e-5 vs 4e-6, restored 3.0e-3 vs 3.6e-3 from the reference -- the
        # sampler chain's own fp32 conditioning dominates), and it costs 5 % throughput (tools/x3_free_running.py)
        self.encoder_fp32_under_x3 = False
        # stage A on the fp32 kernels in the bf16 configuration too (review r5 weak 1.ii: the sampler chain amplifies the encoder's rounding;
        # stage A runs under C + D of the previous batch on the side stream, so its precision costs little time): measured in DESIGN 5
        self.encoder_fp32 = False
        self.encoder_x3 = False      # stage A on the split-precision bf16 kernels (vsp_conv2d_bf16x3: fp32-grade results on the bf16 pipe) under the bf16 configuration
        self.overlap_split = "auto"  # run_batches: which part of stages A + B runs on the side stream (see there); auto = "h" (fp32) / "ab" (bf16 kernels)

    def draw_decode_noise(self, B, image_index0, device):
        """z, prior-decoder, encoder and decoder noise maps of one batch in ONE launch (keyed mode)."""
        import random
        from . import hip_ops as H
        size = self.generator.size
        gen, enc, dec = noise_map_shapes(size, B, gen_size=self.psp.E4Enet.decoder.size if self.with_sample else size)
        # the reference flips ONE coin per batch for style mixing (restoration_test.py:77-82): keyed by the batch's first image
        two = self.mixing > 0 and random.Random(hash((self.noise_seed, int(image_index0)))).random() < self.mixing
        zs = [(B, self.generator.style_dim)] * (2 if two else 1)
        ids = ([H.SEG_Z + i for i in range(len(zs))] + [H.SEG_GEN + i for i in range(len(gen))]
               + [H.SEG_ENC + i for i in range(len(enc))] + [H.SEG_DEC + i for i in range(len(dec))])
        t = H.keyed_fill(zs + gen + enc + dec, ids, self.noise_seed, image_index0, device=device, index_tensor=self._index_tensor)
        a, b, c = len(zs), len(zs) + len(gen), len(zs) + len(gen) + len(enc)
        return t[:a], t[a:b], t[b:c], t[c:]

    @torch.no_grad()
    def encode(self, low_imgs, x_T=None, image_index0=0, handoff=None):
        """Stages A + B: (low_latent, pre_dic_latent).  Small-map convolutions and the latency-bound sampler chain.
        `handoff` (run_batches): an object with .point and .go(tensors) that moves the rest of the call to another stream -- "h": inside the
        encoder, behind its last chip-filling convolution (e4e.Encoder4Editing.forward); "b": between the encoder and the sampler chain."""
        from . import e4e, hip_ops
        owned = False
        if x_T is None and self.noise_seed is not None:
            x_T = hip_ops.keyed_fill([(low_imgs.shape[0], 18, 512)], [hip_ops.SEG_XT], self.noise_seed, image_index0,
                                     device=low_imgs.device, index_tensor=self._index_tensor)[0]
            owned = True
        mode = hip_ops.BF16_CONV
        if mode is True and self.encoder_x3 and not self.encoder_fp32:
            hip_ops.BF16_CONV = "x3"
        elif (mode == "x3" and self.encoder_fp32_under_x3) or (mode and self.encoder_fp32):
            # the sampler chain amplifies a perturbation of its condition ~2000x with random weights: the encoder that
            # feeds it keeps the fp32 kernels, the split-precision kernels serve stages C + D (80 % of the FLOPs)
            hip_ops.BF16_CONV = False
        e4e.HANDOFF = handoff
        try:
            low_latent = self.psp.get_w_plus(low_imgs)
        finally:
            hip_ops.BF16_CONV = mode
            e4e.HANDOFF = None
        if handoff is not None:
            if handoff.point == "b":
                handoff.go([low_latent])
            if x_T is not None:
                handoff.also(x_T)        # drawn on the first stream, read by the chain on the second
        kw = {"x_T_owned": True} if (owned and isinstance(self.diffusion, My_DDPM)) else {}
        pre = self.diffusion(x=low_latent, condi_in=low_latent, training=False, x_T=x_T, **kw)
        return low_latent, pre

    @torch.no_grad()
    def decode(self, low_imgs, low_latent, pre, z=None, gen_noise=None, enc_noise=None, dec_noise=None, inject_index=None,
               image_index0=0):
        """Stages C + D: the StyleGAN2 prior and the restoration network (the compute-bound 97 % of the FLOPs)."""
        noise, gen_noise, enc_noise, dec_noise, inject_index = self._decode_noise(low_imgs, z, gen_noise, enc_noise, dec_noise,
                                                                                   inject_index, image_index0)
        sample, feats = self.prior(pre, gen_noise)
        restored = self.restore(low_imgs, feats, pre, noise, inject_index, enc_noise, dec_noise)
        return {"restored": restored, "style_sample": sample, "latent": low_latent, "pre_latent": pre}

    def _decode_noise(self, low_imgs, z=None, gen_noise=None, enc_noise=None, dec_noise=None, inject_index=None, image_index0=0):
        """The draws of stages C + D (keyed mode: one fill for all of them): (mixing noise list, prior noise, encoder noise, decoder
        noise, inject_index)."""
        B = low_imgs.shape[0]
        if self.noise_seed is not None and (z is None or gen_noise is None or enc_noise is None or dec_noise is None):
            kz, kg, ke, kd = self.draw_decode_noise(B, image_index0, low_imgs.device)
            z = kz if z is None else z
            if len(z) > 1 and inject_index is None:
                import random
                inject_index = random.Random(hash((self.noise_seed, int(image_index0), 1))).randint(1, self.generator.n_latent - 1)
            gen_noise = kg if gen_noise is None else gen_noise
            enc_noise, dec_noise = (ke if enc_noise is None else enc_noise), (kd if dec_noise is None else dec_noise)
        noise = z if z is not None else mixing_noise(B, self.generator.style_dim, self.mixing, low_imgs.device)
        return noise, gen_noise, enc_noise, dec_noise, inject_index

    def _act_mode(self):
        from . import hip_ops
        return bool(self.act_bf16 and hip_ops.BF16_CONV is True)

    @torch.no_grad()
    def prior(self, pre, gen_noise=None):
        """Stage C alone (restoration_test.py:130): (style sample or None, the prior's feature pyramid)."""
        from . import hip_ops
        prev = hip_ops.ACT_BF16
        hip_ops.ACT_BF16 = self._act_mode()
        try:
            return self.psp.get_stylegan_feats(pre, noise=gen_noise, with_sample=self.with_sample)
        finally:
            hip_ops.ACT_BF16 = prev

    @torch.no_grad()
    def restore(self, low_imgs, feats, pre, noise, inject_index=None, enc_noise=None, dec_noise=None):
        """Stage D alone (restoration_test.py:131)."""
        from . import hip_ops
        prev = hip_ops.ACT_BF16
        hip_ops.ACT_BF16 = self._act_mode()
        try:
            return self.generator(low_imgs, feats, pre, noise, inject_index=inject_index, enc_noise=enc_noise, dec_noise=dec_noise)
        finally:
            hip_ops.ACT_BF16 = prev

    @torch.no_grad()
    def __call__(self, low_imgs, z=None, x_T=None, gen_noise=None, enc_noise=None, dec_noise=None, inject_index=None,
                 image_index0=0):
        """low_imgs (B,3,512,512) in [-1,1] on the device -> dict(restored, style_sample, latent, pre_latent).
        All keyword tensors are optional explicit replacements of the reference's RNG draws (parity runs); image_index0 =
        GLOBAL index of the batch's first image (keyed mode)."""
        low_latent, pre = self.encode(low_imgs, x_T=x_T, image_index0=image_index0)
        return self.decode(low_imgs, low_latent, pre, z=z, gen_noise=gen_noise, enc_noise=enc_noise, dec_noise=dec_noise,
                           inject_index=inject_index, image_index0=image_index0)

    @torch.no_grad()
    def run_batches(self, batches):
        """Software-pipelined loop over an iterable of device batches (the `for batch in loader` of restoration_test.py):
        stages A + B of batch i+1 run on a second HIP stream while stages C + D of batch i run on the caller's stream.  A + B
        are small-map / latency-bound work that leaves most CUs idle; overlapped with the big convolutions of the previous
        batch they cost almost nothing.  Yields the same dicts as __call__, in order.  An item is a device batch or a pair
        (batch, image_index0) (keyed mode: GLOBAL index of its first image; plain batches count up from 0)."""
        from . import hip_ops
        main = torch.cuda.current_stream()
        if not hasattr(self, "_side"):
            self._side = _side_stream()
        side = self._side
        # what runs where (round 6, tools/bench_hidden_cost.py: at batch 8 stages C + D alone take 26.3 ms, A + B alone 14.9 ms, and with
        # ALL of A + B on the side stream the step took 39.6 ms -- two streams of chip-filling convolutions take each other's CUs, LDS and
        # L2; only launch- / latency-bound work hides):
        #   "h"  (default) the encoder's trunk and its chip-filling head stages on the MAIN stream in front of C + D of the previous
        #        batch; the small-map head stages, the code assembly and the sampler chain on the side stream underneath C + D
        #   "b"  the whole encoder on the main stream, the chain alone on the side stream
        #   "ab" stages A + B on the side stream (rounds 2-5);  "abc": stage C there too (measured slower in round 5)
        #   "auto" = "h" on the fp32 kernels (same box: 204.2 against 202.2 img/s for "ab"), "ab" in the bf16-kernel configuration, whose stage A
        #        is shorter and less chip-filling (same box, after the DDIM fix: "ab" 530.3 / 530.8, "h" 524.1 / 526.7, "b" 523.2 img/s)
        split_mode = tune_env("VSP_OVERLAP_SPLIT", self.overlap_split)
        if split_mode == "auto":
            split_mode = "ab" if hip_ops.BF16_CONV is True else "h"
        split = split_mode == "abc"

        counter = [0]

        class _Handoff:
            def __init__(self, point):
                self.point, self.done = point, False

            def go(self, tensors):
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                for t in tensors:
                    t.record_stream(side)      # allocated from the main stream's pool, read on the side stream
                torch.cuda.set_stream(side)
                self.done = True

            def also(self, t):
                if self.done:
                    t.record_stream(side)

        def start(item):
            batch, idx0 = item if isinstance(item, (tuple, list)) else (item, counter[0])
            counter[0] = idx0 + batch.shape[0]
            if split_mode in ("h", "b"):
                ho = _Handoff(split_mode)
                try:
                    lat, pre = self.encode(batch, image_index0=idx0, handoff=ho)   # starts on `main`, ends on `side`
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream())
                finally:
                    torch.cuda.set_stream(main)
                for t in _tensors((lat, pre)):
                    t.record_stream(main)      # allocated on the side stream's pool, consumed on the main stream
                return batch, lat, pre, ev, idx0, None
            side.wait_stream(main)  # the batch (and everything enqueued before) is visible to the side stream
            extra = None
            with torch.cuda.stream(side):
                lat, pre = self.encode(batch, image_index0=idx0)
                if split:
                    nz = self._decode_noise(batch, image_index0=idx0)
                    sample, feats = self.prior(pre, nz[1])
                    extra = (nz, sample, feats)
                ev = torch.cuda.Event()
                ev.record(side)
            for t in _tensors((lat, pre, extra)):
                t.record_stream(main)  # allocated on the side stream's pool, consumed on the main stream
            return batch, lat, pre, ev, idx0, extra

        it = iter(batches)
        try:
            cur = start(next(it))
        except StopIteration:
            return
        while cur is not None:
            try:
                nxt = start(next(it))  # enqueue A + B of the next batch BEFORE C + D of this one
            except StopIteration:
                nxt = None
            batch, lat, pre, ev, idx0, extra = cur
            main.wait_event(ev)
            if extra is None:
                yield self.decode(batch, lat, pre, image_index0=idx0)
            else:
                (noise, _g, enc_noise, dec_noise, inject_index), sample, feats = extra
                restored = self.restore(batch, feats, pre, noise, inject_index, enc_noise, dec_noise)
                yield {"restored": restored, "style_sample": sample, "latent": lat, "pre_latent": pre}
            cur = nxt


    # ------------------------------------------------------------------------------------------------------------------
    # hipGraph replay.  One batch is ~1000 kernel launches (about 10 ms of Python + launch work): at batch 1 the host, not the
    # GPU, sets the latency, and at batch 8 the small-map levels of C + D leave bubbles.  Stages A + B and stages C + D are
    # captured ONCE as two HIP graphs (torch.cuda.CUDAGraph: the C ABI enqueues on whatever stream is current, so its launches
    # record like any other; device RNG draws are graph-safe) and replayed per batch; run_batches keeps the two-stream overlap
    # by replaying the A + B graph of batch i+1 on the side stream under the C + D graph of batch i.
    @torch.no_grad()
    def capture_graphs(self, example, x_T=None, z=None, gen_noise=None, enc_noise=None, dec_noise=None):
        """Capture for batches shaped like `example` (B,3,512,512, device).  Needs mixing == 0 (the reference's style mixing
        draws from Python's `random`, which changes the launch sequence from batch to batch).  The optional tensors replace
        the device RNG draws as in __call__; they are captured by reference (static: overwrite their contents between replays
        to change the noise)."""
        if self.mixing != 0:
            raise RuntimeError("capture_graphs: style mixing draws host random numbers per batch; use mixing=0")
        cur = torch.cuda.current_stream()
        warm = torch.cuda.Stream()
        warm.wait_stream(cur)
        with torch.cuda.stream(warm):  # first calls pack weights, raise LDS limits, resolve device constants: not capturable
            for _ in range(2):
                self(example)
        cur.wait_stream(warm)
        torch.cuda.synchronize()
        g = {"e_in": example.clone()}
        if self.noise_seed is not None:
            # keyed draws inside a graph: the kernels read the batch's global image index from device memory (one scalar per
            # graph: A + B of batch i+1 replays while C + D of batch i is still running)
            g["idx_e"] = torch.zeros(1, dtype=torch.int64, device=example.device)
            g["idx_d"] = torch.zeros(1, dtype=torch.int64, device=example.device)
        try:
            self._index_tensor = g.get("idx_e")
            g["E"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g["E"]):
                g["e_lat"], g["e_pre"] = self.encode(g["e_in"], x_T=x_T)
            g["d_in"], g["d_lat"], g["d_pre"] = example.clone(), g["e_lat"].clone(), g["e_pre"].clone()
            self._index_tensor = g.get("idx_d")
            g["D"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g["D"]):
                g["out"] = self.decode(g["d_in"], g["d_lat"], g["d_pre"], z=z, gen_noise=gen_noise, enc_noise=enc_noise,
                                       dec_noise=dec_noise)
        finally:
            self._index_tensor = None
        self._graphs = g
        return self

    @torch.no_grad()
    def run_batches_graphed(self, batches):
        """run_batches over the captured graphs.  The yielded dict holds the graphs' STATIC output tensors: consume (or copy)
        them before asking for the next batch."""
        g = self._graphs
        main = torch.cuda.current_stream()
        if not hasattr(self, "_side"):
            self._side = _side_stream()
        side = self._side

        counter = [0]

        def start(item, after):
            """A + B of `batch` on the side stream, once `after` (the main stream's copy of the previous latents) is done."""
            batch, idx0 = item if isinstance(item, (tuple, list)) else (item, counter[0])
            counter[0] = idx0 + batch.shape[0]
            if after is not None:
                side.wait_event(after)
            else:
                side.wait_stream(main)
            with torch.cuda.stream(side):
                g["e_in"].copy_(batch)
                if "idx_e" in g:
                    g["idx_e"].fill_(idx0)
                g["E"].replay()
                ev = torch.cuda.Event()
                ev.record(side)
            return batch, ev, idx0

        it = iter(batches)
        try:
            cur = start(next(it), None)
        except StopIteration:
            return
        while cur is not None:
            batch, ev, idx0 = cur
            main.wait_event(ev)
            if "idx_d" in g:
                g["idx_d"].fill_(idx0)
            g["d_in"].copy_(batch)
            g["d_lat"].copy_(g["e_lat"])
            g["d_pre"].copy_(g["e_pre"])
            copied = torch.cuda.Event()
            copied.record(main)
            try:
                nxt = start(next(it), copied)  # enqueue A + B of the next batch BEFORE C + D of this one
            except StopIteration:
                nxt = None
            g["D"].replay()
            yield g["out"]
            cur = nxt


def gather_restored(local, counts=None):
    """All-gather the per-rank restored images (B_r,3,H,W) into the full batch on every rank (RCCL all-gather over
    xGMI; gloo on CPU in tests).  `counts` = per-rank batch sizes when the split is ragged."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    if counts is None or len(set(counts)) == 1:
        out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    mx = max(counts)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    pad[:local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], 0)


class RestoredGather:
    """The all-gather of the restored images as an ASYNCHRONOUS, preallocated exchange (round 3): `start(local)` enqueues the
    collective behind everything already on the current stream and returns at once (RCCL runs it on the communicator's own
    stream); `result()` makes the current stream wait for it and hands out the full batch.  Called as

        for out in pipe.run_batches(batches):          # C + D of batch i enqueued ...
            h = gather.start(out["restored"])          # ... its all-gather behind them ...
            if pending is not None: full = pending.result()   # ... and the PREVIOUS batch's result is waited for only now, i.e.
            pending = h                                 #     its exchange ran under this batch's C + D

    the exchange of batch i overlaps the convolutions of batch i+1 instead of sitting on the critical path after each batch, and no
    (W B, 3, 512, 512) tensor is allocated per step: two output buffers alternate (a handle's buffer is reused two starts later --
    consume or copy the result before that).  Ragged splits (`counts` = per-rank batch sizes, equal on every rank) pad every
    rank's share to the largest one in a preallocated staging tensor and return a view of the gathered rows in rank order.
    Single-rank / uninitialised process group: `result()` returns `local` unchanged.

    Lifetime of `local`: the collective READS it asynchronously, so it must not be overwritten before `result()` of that handle.  A
    fresh tensor per batch (the eager paths) is kept alive by the handle and is safe.  A buffer that the caller REUSES for the next batch
    -- the static output of a captured graph (`run_batches_graphed`) -- must be passed with `stage=True`: `start` then copies it on the
    current stream into a staging buffer this object owns (one per slot) before the collective is enqueued, so the next replay may
    overwrite `local` at once (without it a rank that runs ahead sends batch i+1's pixels into gather i)."""

    def __init__(self, group=None):
        self.group = group
        self._out = {}      # (slot, shape key) -> gathered buffer
        self._pad = {}      # (slot, shape key) -> padded staging buffer of this rank
        self._n = 0

    class Handle:
        def __init__(self, work, out, counts, local, mx):
            self.work, self.out, self.counts, self.local, self.mx = work, out, counts, local, mx

        def result(self):
            if self.work is None:
                return self.local
            self.work.wait()          # orders the CURRENT stream behind the collective (no host block on RCCL)
            if self.counts is None:
                return self.out
            if all(c == self.mx for c in self.counts):
                return self.out
            rows = self.out.view((len(self.counts), self.mx) + tuple(self.out.shape[1:]))
            return torch.cat([rows[r, :c] for r, c in enumerate(self.counts)], 0)

    def start(self, local, counts=None, stage=False):
        import torch.distributed as dist
        if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return RestoredGather.Handle(None, None, None, local, 0)
        world = dist.get_world_size(self.group)
        slot = self._n & 1
        self._n += 1
        mx = local.shape[0] if counts is None else max(counts)
        key = (slot, mx, tuple(local.shape[1:]), local.dtype, local.device)
        out = self._out.get(key)
        if out is None:
            out = self._out[key] = torch.empty((world * mx,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
        src = local.contiguous()
        if src.shape[0] != mx or stage:  # a shorter share than the longest one, or a buffer the caller will overwrite: owned staging copy
            pad = self._pad.get(key)
            if pad is None:
                pad = self._pad[key] = torch.zeros((mx,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
            pad[:src.shape[0]].copy_(src)
            src = pad
        work = dist.all_gather_into_tensor(out, src, group=self.group, async_op=True)
        return RestoredGather.Handle(work, out, counts, local, mx)
