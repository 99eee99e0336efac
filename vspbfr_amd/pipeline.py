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

from .diffusion import Code_diffuser, My_DDPM
from .e4e import E4e_embedding
from .restorenet import Restoration_net, mixing_noise


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


class RestorationPipeline:
    def __init__(self, generator: Restoration_net, psp_embedding: E4e_embedding, diffusion: My_DDPM, mixing=0.5,
                 with_sample=True):
        self.generator, self.psp, self.diffusion = generator.eval(), psp_embedding.eval(), diffusion.eval()
        self.mixing, self.with_sample = mixing, with_sample
        # option for the split-precision configuration: keep the encoder on the fp32 kernels.  Off: measured on the pinned case the
        # free-running result is the same either way (codes 2.5e-5 vs 4e-6, restored 3.0e-3 vs 3.6e-3 from the reference -- the
        # sampler chain's own fp32 conditioning dominates), and it costs 5 % throughput (tools/x3_free_running.py)
        self.encoder_fp32_under_x3 = False

    @torch.no_grad()
    def encode(self, low_imgs, x_T=None):
        """Stages A + B: (low_latent, pre_dic_latent).  Small-map convolutions and the latency-bound sampler chain."""
        from . import hip_ops
        mode = hip_ops.BF16_CONV
        if mode == "x3" and self.encoder_fp32_under_x3:
            # the sampler chain amplifies a perturbation of its condition ~2000x with random weights: the encoder that
            # feeds it keeps the fp32 kernels, the split-precision kernels serve stages C + D (80 % of the FLOPs)
            hip_ops.BF16_CONV = False
        try:
            low_latent = self.psp.get_w_plus(low_imgs)
        finally:
            hip_ops.BF16_CONV = mode
        pre = self.diffusion(x=low_latent, condi_in=low_latent, training=False, x_T=x_T)
        return low_latent, pre

    @torch.no_grad()
    def decode(self, low_imgs, low_latent, pre, z=None, gen_noise=None, enc_noise=None, dec_noise=None, inject_index=None):
        """Stages C + D: the StyleGAN2 prior and the restoration network (the compute-bound 97 % of the FLOPs)."""
        B = low_imgs.shape[0]
        noise = z if z is not None else mixing_noise(B, self.generator.style_dim, self.mixing, low_imgs.device)
        sample, feats = self.psp.get_stylegan_feats(pre, noise=gen_noise, with_sample=self.with_sample)
        restored = self.generator(low_imgs, feats, pre, noise, inject_index=inject_index, enc_noise=enc_noise,
                                  dec_noise=dec_noise)
        return {"restored": restored, "style_sample": sample, "latent": low_latent, "pre_latent": pre}

    @torch.no_grad()
    def __call__(self, low_imgs, z=None, x_T=None, gen_noise=None, enc_noise=None, dec_noise=None, inject_index=None):
        """low_imgs (B,3,512,512) in [-1,1] on the device -> dict(restored, style_sample, latent, pre_latent).
        All keyword tensors are optional explicit replacements of the reference's RNG draws (parity runs)."""
        low_latent, pre = self.encode(low_imgs, x_T=x_T)
        return self.decode(low_imgs, low_latent, pre, z=z, gen_noise=gen_noise, enc_noise=enc_noise, dec_noise=dec_noise,
                           inject_index=inject_index)

    @torch.no_grad()
    def run_batches(self, batches):
        """Software-pipelined loop over an iterable of device batches (the `for batch in loader` of restoration_test.py):
        stages A + B of batch i+1 run on a second HIP stream while stages C + D of batch i run on the caller's stream.  A + B
        are small-map / latency-bound work that leaves most CUs idle; overlapped with the big convolutions of the previous
        batch they cost almost nothing.  Yields the same dicts as __call__, in order."""
        main = torch.cuda.current_stream()
        if not hasattr(self, "_side"):
            self._side = torch.cuda.Stream()
        side = self._side

        def start(batch):
            side.wait_stream(main)  # the batch (and everything enqueued before) is visible to the side stream
            with torch.cuda.stream(side):
                lat, pre = self.encode(batch)
                ev = torch.cuda.Event()
                ev.record(side)
            for t in (lat, pre):
                t.record_stream(main)  # allocated on the side stream's pool, consumed on the main stream
            return batch, lat, pre, ev

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
            batch, lat, pre, ev = cur
            main.wait_event(ev)
            yield self.decode(batch, lat, pre)
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
        g["E"] = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g["E"]):
            g["e_lat"], g["e_pre"] = self.encode(g["e_in"], x_T=x_T)
        g["d_in"], g["d_lat"], g["d_pre"] = example.clone(), g["e_lat"].clone(), g["e_pre"].clone()
        g["D"] = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g["D"]):
            g["out"] = self.decode(g["d_in"], g["d_lat"], g["d_pre"], z=z, gen_noise=gen_noise, enc_noise=enc_noise,
                                   dec_noise=dec_noise)
        self._graphs = g
        return self

    @torch.no_grad()
    def run_batches_graphed(self, batches):
        """run_batches over the captured graphs.  The yielded dict holds the graphs' STATIC output tensors: consume (or copy)
        them before asking for the next batch."""
        g = self._graphs
        main = torch.cuda.current_stream()
        if not hasattr(self, "_side"):
            self._side = torch.cuda.Stream()
        side = self._side

        def start(batch, after):
            """A + B of `batch` on the side stream, once `after` (the main stream's copy of the previous latents) is done."""
            if after is not None:
                side.wait_event(after)
            else:
                side.wait_stream(main)
            with torch.cuda.stream(side):
                g["e_in"].copy_(batch)
                g["E"].replay()
                ev = torch.cuda.Event()
                ev.record(side)
            return batch, ev

        it = iter(batches)
        try:
            cur = start(next(it), None)
        except StopIteration:
            return
        while cur is not None:
            batch, ev = cur
            main.wait_event(ev)
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
