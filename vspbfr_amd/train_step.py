"""One iteration of the restoration GAN's training loop (reference restoration_train.py:153-255, optimisers :397-408) over the gfx950
operators, data-parallel with a bucketed gradient all-reduce (SURVEY 8f row 2 / BASELINE configs[4]).

    D step        d_logistic(D(real), D(G(low).detach()))                              every iteration
    D regulariser r1 / 2 * R1(D, real) * d_reg_every  (double backward)                every d_reg_every iterations
    G step        g_nonsaturating(D(G(low))) [+ percept_w * percept(fake, real)] [+ id_w * id(fake, real)]
    EMA           g_ema <- decay * g_ema + (1 - decay) * G,  decay = 0.5 ** (32 / 10000)

The frozen front of the path (stage A encoder, stage B sampler, stage C prior) runs through the INFERENCE kernels under no_grad
exactly as in restoration_test; the generator runs `training.restoration_net_forward` (differentiable), the discriminator
`discriminator.Discriminator`.  The perceptual (LPIPS-VGG) and identity (ArcFace) terms of the reference need their pretrained
networks (my_lpips/, Loss/id_loss.py: out of scope): they enter as optional callables with weight 0 by default.

Data parallelism: one process per GPU, every rank owns the same parameters and its shard of the batch; the gradients are averaged
by `OverlappedGradientReducer` (started from inside backward) / `allreduce_gradients` (after it): parameters are packed into flat buckets in REVERSE registration order (the
order backward produces them), each bucket is one asynchronous all-reduce (RCCL over xGMI through torch.distributed "nccl"; gloo
in the CPU tests), unpacked after the last wait.  xGMI is point-to-point -- a ring all-reduce moves 2 (N-1)/N of a bucket over
the slowest link -- so buckets are large (64 MB default: the generator's 450 MB of fp32 gradients = 7 collectives) rather than the
25 MB that suits NVSwitch; parameters without a gradient (find_unused_parameters in the reference's DDP) contribute zeros so that
every rank issues identical collectives."""
import random

import torch

from . import training
from .discriminator import accumulate, d_logistic_loss, d_r1_loss, first_order, g_nonsaturating_loss
from .restorenet import mixing_noise


def allreduce_gradients(params, bucket_bytes=64 << 20, group=None):
    """Average .grad over the ranks of `group` in flat buckets; a no-op without an initialised process group or with one rank."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    world = dist.get_world_size(group)
    params = [p for p in params if p.requires_grad]
    buckets, cur, size = [], [], 0
    for p in reversed(params):
        cur.append(p)
        size += p.numel() * p.element_size()
        if size >= bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
    if cur:
        buckets.append(cur)
    pending = []
    for bucket in buckets:
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        pending.append((bucket, flat, dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)))
    for bucket, flat, work in pending:
        work.wait()
        flat.div_(world)
        off = 0
        for p in bucket:
            n = p.numel()
            if p.grad is None:
                p.grad = flat[off:off + n].view_as(p).clone()
            else:
                p.grad.copy_(flat[off:off + n].view_as(p))
            off += n
    return len(buckets)


class OverlappedGradientReducer:
    """`allreduce_gradients` started from INSIDE the backward pass: a post-accumulate hook per parameter counts a bucket's gradients
    in, and the moment the last one lands the bucket is packed and its all-reduce enqueued (asynchronous: on RCCL the collective runs
    on the communicator's own stream, ordered after the packing, while the compute stream carries on with the rest of backward) --
    the exchange of the 64 MB buckets of the late layers hides under the backward of the early ones.  Same buckets (reverse
    registration order), same averaging and the same zero-filling of parameters without a gradient as `allreduce_gradients`, with
    which it agrees bit for bit (tests/test_distributed_cpu.py).  Buckets become ready in autograd's order, which is the same on every
    rank for the same graph; whatever is incomplete when backward returns is launched by `finish()` in bucket order.

        red = OverlappedGradientReducer(list(net.parameters()))
        with red:                    # arms the hooks; a no-op without an initialised process group or with one rank
            loss.backward()
        # on exit: remaining buckets launched, all waited for, gradients averaged in place"""

    def __init__(self, params, bucket_bytes=64 << 20, group=None):
        self.group = group
        params = list(params)
        # The bucket plan is fixed here, and every rank must issue the SAME collectives: a parameter that is frozen at this moment
        # (a generator handed over from an inference pipeline after requires_grad_(False), a discriminator another trainer left
        # frozen) would silently drop out of the exchange while backward still computes its gradient once it is thawed -- ranks
        # would diverge.  So: every parameter is bucketed and hooked; which of them receive a gradient in a given backward is
        # decided per pass (`finish` zero-fills the rest), and a caller that really wants a subset passes exactly that subset.
        frozen = [p for p in params if not p.requires_grad]
        if frozen and len(frozen) < len(params):
            raise ValueError(f"OverlappedGradientReducer: {len(frozen)} of {len(params)} parameters are frozen at construction; pass the "
                             "trainable subset explicitly or thaw the module first (requires_grad_(True))")
        # an entirely frozen module (its owner toggles requires_grad per step): a hook can only be registered on a grad-requiring leaf,
        # and it survives later toggles -- so the flag is raised for the registration only and put back: the caller's module is not
        # silently thawed (a generator shared with an inference pipeline stays frozen until its owner thaws it)
        for p in frozen:
            p.requires_grad_(True)
        self.params = params
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * p.element_size()
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self.bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self.active = False
        self.launched = 0
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._hook)
        for p in frozen:
            p.requires_grad_(False)

    def _world(self):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        return dist.get_world_size(self.group)

    def _hook(self, p):
        if not self.active:
            return
        i = self.bucket_of[id(p)]
        self.count[i] += 1
        if self.count[i] == len(self.buckets[i]):
            self._launch(i)

    def _launch(self, i):
        import torch.distributed as dist
        bucket = self.buckets[i]
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in bucket])
        self.pending[i] = (flat, dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self.launched += 1

    def __enter__(self):
        self.world = self._world()
        self.active = self.world > 1
        self.count = [0] * len(self.buckets)
        self.pending = {}
        self.launched = 0
        return self

    def __exit__(self, exc_type, exc, tb):
        active, self.active = self.active, False
        if not active or exc_type is not None:
            return False
        for i in range(len(self.buckets)):          # buckets some parameter of which never received a gradient
            if i not in self.pending:
                self._launch(i)
        for i, bucket in enumerate(self.buckets):
            flat, work = self.pending[i]
            work.wait()
            flat.div_(self.world)
            off = 0
            for p in bucket:
                n = p.numel()
                if p.grad is None:
                    p.grad = flat[off:off + n].view_as(p).clone()
                else:
                    p.grad.copy_(flat[off:off + n].view_as(p))
                off += n
        self.pending = {}
        return False


def requires_grad(model, flag=True):
    for p in model.parameters():
        p.requires_grad_(flag)


class RestorationTrainer:
    def __init__(self, generator, g_ema, discriminator, psp_embedding=None, diffusion=None, lr=0.002, g_reg_every=4, d_reg_every=16,
                 r1=10.0, mixing=0.9, percept_loss=None, percept_weight=0.0, id_loss=None, id_weight=0.0, bucket_bytes=64 << 20,
                 augment=False, augment_p=0.0, ada_target=0.6, ada_length=500 * 1000, ada_every=8):
        self.G, self.G_ema, self.D = generator, g_ema, discriminator
        self.psp, self.diffusion = psp_embedding, diffusion
        self.d_reg_every, self.r1, self.mixing = d_reg_every, r1, mixing
        self.percept_loss, self.percept_weight, self.id_loss, self.id_weight = percept_loss, percept_weight, id_loss, id_weight
        self.bucket_bytes = bucket_bytes
        # ADA (restoration_train.py:144-148, 173-180, 194-196): fixed probability augment_p, or adaptive when it is 0
        self.augment_on, self.ada_aug_p, self.ada = augment, augment_p, None
        if augment and augment_p == 0:
            from .non_leaking import AdaptiveAugment
            self.ada = AdaptiveAugment(ada_target, ada_length, ada_every, next(discriminator.parameters()).device)
        g_ratio, d_ratio = g_reg_every / (g_reg_every + 1), d_reg_every / (d_reg_every + 1)   # restoration_train.py:397-408
        # device parameters: the single-kernel (fused) multi-tensor Adam; same update rule as restoration_train.py:123-131
        fused = all(p.is_cuda for p in generator.parameters()) and all(p.is_cuda for p in discriminator.parameters())
        self.g_optim = torch.optim.Adam(generator.parameters(), lr=lr * g_ratio, betas=(0 ** g_ratio, 0.99 ** g_ratio), fused=fused)
        self.d_optim = torch.optim.Adam(discriminator.parameters(), lr=lr * d_ratio, betas=(0 ** d_ratio, 0.99 ** d_ratio), fused=fused)
        self.accum = 0.5 ** (32 / (10 * 1000))
        accumulate(g_ema, generator, 0)
        # the step toggles requires_grad on G and D every iteration and leaves D frozen at its end; a trainer built on modules in that
        # state (or on a generator taken from an inference pipeline) must still exchange every gradient
        requires_grad(generator, True)
        requires_grad(discriminator, True)
        self.generator_bytes = sum(p.numel() * p.element_size() for p in generator.parameters())
        self.g_reducer = OverlappedGradientReducer(list(generator.parameters()), bucket_bytes)
        self.d_reducer = OverlappedGradientReducer(list(discriminator.parameters()), bucket_bytes)

    @torch.no_grad()
    def front(self, low_img):
        """Stages A, B, C of the path (frozen): W+ codes -> denoised latent -> prior features (restoration_train.py:167-171)."""
        low_latent = self.psp.get_w_plus(low_img)
        latent = self.diffusion(x=low_latent, condi_in=low_latent, training=False)
        _, de_feats = self.psp.get_stylegan_feats(latent, with_sample=False)
        return de_feats, latent

    def generate(self, low_img, de_feats, latent, noise, enc_noise=None, dec_noise=None):
        B, size = low_img.shape[0], self.G.size
        if enc_noise is None:
            from .pipeline import noise_map_shapes
            _, es, ds = noise_map_shapes(size, B)
            enc_noise = [torch.randn(s, device=low_img.device) for s in es]
            dec_noise = [torch.randn(s, device=low_img.device) for s in ds]
        inject = None if len(noise) < 2 else random.randint(1, self.G.n_latent - 1)
        if not torch.is_grad_enabled():   # the discriminator step's fake batch: the fused inference forward (same parameters, same mode)
            return self.G(low_img, de_feats, latent, noise, inject_index=inject, enc_noise=enc_noise, dec_noise=dec_noise)
        return training.restoration_net_forward(self.G, low_img, de_feats, latent, noise, enc_noise, dec_noise, inject_index=inject)

    def _aug(self, img):
        if not self.augment_on:
            return img
        from .non_leaking import augment
        return augment(img, self.ada_aug_p)[0]

    def step(self, i, low_img, real_img, de_feats=None, latent=None):
        """Iteration i on this rank's shard (images in [-1, 1] on the device).  de_feats / latent: precomputed outputs of the
        frozen front (tests); default = run it.  Returns the loss dict of the reference's logger."""
        dev, B = low_img.device, low_img.shape[0]
        if de_feats is None:
            de_feats, latent = self.front(low_img)
        de_feats = [f.detach().float() if f.dtype != torch.float32 else f.detach() for f in de_feats]
        latent = latent.detach()
        losses = {}
        # ---- discriminator
        requires_grad(self.G, False)
        requires_grad(self.D, True)
        with torch.no_grad():
            fake = self.generate(low_img, de_feats, latent, mixing_noise(B, self.G.style_dim, self.mixing, dev))
        with first_order():   # logistic loss: first-order passes (the R1 pass below keeps the twice-differentiable operators)
            fake_pred, real_pred = self.D(self._aug(fake.detach())), self.D(self._aug(real_img.detach().clone()))
        d_loss = d_logistic_loss(real_pred, fake_pred)
        self.D.zero_grad(set_to_none=True)
        with self.d_reducer:
            d_loss.backward()
        self.d_optim.step()
        losses.update(d=d_loss.detach(), real_score=real_pred.mean().detach(), fake_score=fake_pred.mean().detach())
        if self.ada is not None:
            self.ada_aug_p = self.ada.tune(real_pred)
        if i % self.d_reg_every == 0:
            x = real_img.detach().clone().requires_grad_(True)
            pred = self.D(self._aug(x))
            r1_loss = d_r1_loss(pred, x)
            self.D.zero_grad(set_to_none=True)
            with self.d_reducer:
                (self.r1 / 2 * r1_loss * self.d_reg_every + 0 * pred[0]).backward()
            self.d_optim.step()
            losses["r1"] = r1_loss.detach()
        # ---- generator
        requires_grad(self.G, True)
        requires_grad(self.D, False)
        fake = self.generate(low_img, de_feats, latent, mixing_noise(B, self.G.style_dim, self.mixing, dev))
        with first_order():
            g_loss = g_nonsaturating_loss(self.D(self._aug(fake)))
        losses["g"] = g_loss.detach()
        if self.percept_loss is not None and self.percept_weight > 0:
            t = self.percept_loss(fake, real_img.detach()).sum() * self.percept_weight
            losses["g_percept_loss"], g_loss = t.detach(), g_loss + t
        if self.id_loss is not None and self.id_weight > 0:
            t = self.id_loss(fake, real_img.detach()) * self.id_weight
            losses["g_id_loss"], g_loss = t.detach(), g_loss + t
        self.G.zero_grad(set_to_none=True)
        with self.g_reducer:
            g_loss.backward()
        losses["grad_buckets"] = self.g_reducer.launched
        self.g_optim.step()
        accumulate(self.G_ema, self.G, self.accum)
        return losses



class CodeDiffuserTrainer:
    """One iteration of the stage-B training loop (reference code_diffuser_train.py:153-190, optimiser :307-311): the Code_diffuser
    learns to map the e4e codes of the degraded image to those of the clean one, through the training-mode sampler
    (q_sample at T - 1, T posterior-mean steps) and -- for the perceptual / identity terms -- through the frozen StyleGAN2 prior.

        latent_loss = L1(pred_IPR_list[-1], target codes)            (KDLoss's second output; the KL term is only logged)
                      + 0.1 * LPIPS(prior(pred codes), real).mean() + 0.1 * ID(prior(pred codes), real)

    Encoder, prior and loss networks are frozen; gradients are averaged over the ranks with `allreduce_gradients` (72 tensors,
    17 MB: one bucket)."""

    def __init__(self, diffusion, psp_embedding, lr=0.002, g_reg_every=4, percept_loss=None, percept_weight=0.5, id_loss=None,
                 id_weight=0.1, bucket_bytes=64 << 20):
        self.diffusion, self.psp = diffusion, psp_embedding
        self.percept_loss, self.percept_weight, self.id_loss, self.id_weight = percept_loss, percept_weight, id_loss, id_weight
        self.bucket_bytes = bucket_bytes
        self.cri_kd = training.KDLoss()
        ratio = g_reg_every / (g_reg_every + 1)
        self.params = list(diffusion.model.parameters())
        for p in self.params:
            p.requires_grad_(True)
        self.optim = torch.optim.Adam(self.params, lr=lr * ratio, betas=(0 ** ratio, 0.99 ** ratio), fused=all(p.is_cuda for p in self.params))
        self.reducer = OverlappedGradientReducer(self.params, bucket_bytes)

    def step(self, low_img, real_img, low_latent=None, target=None, q_noise=None, gen_noise=None):
        """low_img / real_img in [-1, 1] on the device.  low_latent / target: precomputed codes (tests); q_noise / gen_noise: the
        random draws (tests).  Returns the loss dict of the reference's logger."""
        with torch.no_grad():
            if low_latent is None:
                low_latent = self.psp.get_w_plus(low_img)
            if target is None:
                target = self.psp.get_w_plus(real_img)
        pred, seq = training.ddpm_training_forward(self.diffusion, low_latent.detach(), low_latent.detach(), q_noise)
        l_kd, l_abs = self.cri_kd([target.detach()], [seq[-1]])
        loss = l_abs
        losses = {"latent_loss": l_abs.detach(), "l_kd": l_kd.detach()}
        use_p = self.percept_loss is not None and self.percept_weight > 0
        use_i = self.id_loss is not None and self.id_weight > 0
        if use_p or use_i:
            restore = self.psp.get_stylegan_featsV2(pred, grad=True, return_feat=False, noise=gen_noise)
            if use_p:
                t = self.percept_loss(restore, real_img.detach()).mean() * 0.1
                losses["latent_percept_loss"], loss = t.detach(), loss + t
            if use_i:
                t = self.id_loss(restore, real_img.detach()) * 0.1
                losses["latent_id_loss"], loss = t.detach(), loss + t
        for p in self.params:
            p.grad = None
        with self.reducer:
            loss.backward()
        self.optim.step()
        losses["pred_latent"] = pred.detach()
        return losses
