"""Code_diffuser (4 x TACC_block on W+ latents) and the My_DDPM sampler on gfx950 -- constructor/forward API and
state-dict keys of the reference's models/CodeDiffuser.py:63-140 and ldm/ddpm.py:253-429.

Execution (MI355X-first): every F.linear / torch.matmul is the strided fp32-MFMA small GEMM of the C ABI; the
conditioning vector c = [embd, t/T] never materialises as a 513-wide tensor on the hot path: a Linear(513 -> 512) is
split into its 512-wide part (one GEMM on the step-independent `embd`, hoisted out of the sampling loop by
`My_DDPM.forward`) plus t/T times its last weight column.  Everything that depends only on (embd, t) -- Q, k, gamma,
beta of every block -- is therefore computed once per (condition, t) outside the x-dependent chain.
"""
import math
from functools import partial

import numpy as np
import torch
from torch import nn

from . import hip_ops as H

SQRT2 = math.sqrt(2.0)


class spatial_attention(nn.Module):
    def __init__(self, in_dim=18, latent_dim=512):
        super().__init__()
        self.q_matrix = nn.Linear(latent_dim, latent_dim, bias=False)
        self.k_matrix = nn.Linear(latent_dim + 1, latent_dim, bias=False)
        self.v_matrix = nn.Linear(latent_dim, latent_dim, bias=False)
        self.dk = latent_dim


class _Head(nn.Sequential):
    """Linear(D+1, D) -> LayerNorm -> ScaledLeakyReLU -> Linear(D, D) -> Sigmoid | ScaledLeakyReLU (keys .0 .1 .3)."""

    def __init__(self, dim, last):
        super().__init__(nn.Linear(dim + 1, dim), nn.LayerNorm([dim]), nn.Identity(), nn.Linear(dim, dim), nn.Identity())
        self.last = last  # 2 = sigmoid, 1 = leaky-relu * sqrt2


def _lin513(lin, embd_part, tfrac):
    """Linear(513 -> 512)(cat[embd, t/T]) given embd_part = embd @ W[:, :512]^T (+bias): adds t/T * W[:, 512]."""
    return embd_part + tfrac * lin.weight[:, -1]


class TACC_block(nn.Module):
    def __init__(self, latent_dim=512, in_dim=18):
        super().__init__()
        d = latent_dim
        self.dim = d
        self.q_matrix = nn.Linear(d + 1, d, bias=False)
        self.k_matrix = nn.Linear(d, d, bias=False)
        self.v_matrix = nn.Linear(d, d, bias=False)
        self.gamma_ = _Head(d, 2)
        self.beta_ = _Head(d, 1)
        self.attention_layer = spatial_attention(in_dim=in_dim, latent_dim=d)
        self.dk = 18

    # ---- step-independent part: the four Linear(513) layers applied to embd only
    def embed(self, embd):
        d = self.dim
        rows = embd.reshape(-1, d)

        def proj(lin):
            # the first d columns of the (d, d+1) weight as a CONTIGUOUS matrix, cached per weight version: read in place through the
            # row pitch d+1 the rows are only 4-byte aligned and the GEMM falls back to scalar fragment loads (33.7 us per launch
            # against 7 us; 16 launches per batch)
            wv = lin.weight
            stamp = (wv.data_ptr(), wv._version)
            hit = self.__dict__.setdefault("_wsq_cache", {}).get(id(lin))
            if hit is None or hit[0] != stamp:
                hit = (stamp, wv.detach()[:, :d].contiguous())
                self.__dict__["_wsq_cache"][id(lin)] = hit
            out = H.gemm_nt(rows, hit[1], bias=lin.bias)
            return out.view(*embd.shape[:-1], d)
        return {"Q": proj(self.q_matrix), "k": proj(self.attention_layer.k_matrix), "g": proj(self.gamma_[0]),
                "b": proj(self.beta_[0])}

    # ---- (embd, t)-dependent, x-independent part
    def condition(self, emb, tfrac):
        Q = _lin513(self.q_matrix, emb["Q"], tfrac)
        k2 = _lin513(self.attention_layer.k_matrix, emb["k"], tfrac)

        def head(seq, e):
            g = H.layernorm(_lin513(seq[0], e, tfrac).contiguous(), gamma=seq[1].weight, beta=seq[1].bias, post_lrelu=True)
            return H.linear(g, seq[3].weight, seq[3].bias, act=seq.last)
        return {"Q": Q.contiguous(), "k": k2.contiguous(), "gamma": head(self.gamma_, emb["g"]), "beta": head(self.beta_, emb["b"])}

    # ---- x-dependent chain
    def run(self, x, cond):
        B, T, d = x.shape
        xn = H.pixelnorm_dim1(x)                                     # over the 18 tokens (models/CodeDiffuser.py:11-12,92)
        att = self.attention_layer
        K = H.linear(xn, self.k_matrix.weight)
        V = H.linear(xn, self.v_matrix.weight)
        score = H.softmax_lastdim(H.gemm_nt(K, cond["Q"], alpha=1 / math.sqrt(self.dk)))           # (B,18,18)
        h = H.gemm_nt(score, V, dims=(B, T, d, T), a_strides=(T * T, T, 1), b_strides=(T * d, 1, d))  # score @ V
        q2 = H.linear(xn, att.q_matrix.weight)
        v2 = H.linear(xn, att.v_matrix.weight)
        # channel attention: A = softmax_dim1(k^T q / sqrt(d)) (d x d per sample), t = LN(v A)
        A = H.gemm_nt(cond["k"], q2, dims=(B, d, d, T), a_strides=(T * d, 1, d), b_strides=(T * d, 1, d),
                      alpha=1 / math.sqrt(att.dk))
        A = H.softmax_dim1(A)
        t = H.gemm_nt(v2, A, dims=(B, T, d, d), a_strides=(T * d, d, 1), b_strides=(d * d, 1, d))
        t = H.layernorm(t)
        h = H.layernorm(h, add=t)
        return H.film(h, cond["gamma"], cond["beta"])

    def forward(self, x, embd, step):
        tfrac = step[..., :1] if step.dim() == 3 else step
        return self.run(x.contiguous(), self.condition(self.embed(embd.contiguous()), tfrac))


class Code_diffuser(nn.Module):
    def __init__(self, timesteps, dim=512):
        super().__init__()
        self.max_period = timesteps
        self.att_mapper = nn.ModuleList([TACC_block(latent_dim=dim) for _ in range(4)])

    def embed(self, embd):
        return [blk.embed(embd) for blk in self.att_mapper]

    def condition(self, emb, t, n_tokens):
        tfrac = (t.float() / self.max_period).view(-1, 1, 1).expand(-1, n_tokens, 1)
        return [blk.condition(e, tfrac) for blk, e in zip(self.att_mapper, emb)]

    def run(self, x, conds):
        for blk, c in zip(self.att_mapper, conds):
            x = blk.run(x, c)
        return x

    @torch.no_grad()
    def forward(self, x, embd, t):
        embd = embd.contiguous()
        return self.run(x.contiguous(), self.condition(self.embed(embd), t, embd.shape[1]))

    # ---- fused sampler path (all samples share t): `prepare_chain` + ONE vsp_tacc_chain_f32 call for the whole sampler
    # (csrc/tacc_chain.hip: 3 launches per block and step, enqueued from C); `chain_step` is the per-step form of the
    # same computation on the per-launch entry points (csrc/tacc.hip), kept for callers that drive the loop themselves
    def _wcat(self, blk):
        srcs = [blk.k_matrix.weight, blk.v_matrix.weight, blk.attention_layer.q_matrix.weight, blk.attention_layer.v_matrix.weight]
        store = self.__dict__.setdefault("_wcat_cache", {})
        stamp = tuple((w.data_ptr(), w._version) for w in srcs)
        hit = store.get(id(blk))
        if hit is None or hit[0] != stamp:
            wc = torch.cat([w.detach() for w in srcs], 0).contiguous()
            # the same matrix in MFMA fragment order (include/vspbfr_hip.h vsp_tacc_block.wcat_frag): [n / 16][k / 16][k % 16 / 4][n % 16][k % 4]
            n, k = wc.shape
            frag = wc.view(n // 16, 16, k // 16, 4, 4).permute(0, 2, 3, 1, 4).contiguous() if n % 16 == 0 and k % 16 == 0 else None
            hit = (stamp, wc, frag)
            store[id(blk)] = hit
        return hit[1]

    def _wcat_frag(self, blk):
        self._wcat(blk)
        return self.__dict__["_wcat_cache"][id(blk)][2]

    def _tcols(self, blk):
        """The t-columns of the two Linear(513) layers as contiguous vectors (constant: cached like the concatenated matrix)."""
        srcs = [blk.q_matrix.weight, blk.attention_layer.k_matrix.weight]
        store = self.__dict__.setdefault("_tcol_cache", {})
        stamp = tuple((w.data_ptr(), w._version) for w in srcs)
        hit = store.get(id(blk))
        if hit is None or hit[0] != stamp:
            hit = (stamp, tuple(w.detach()[:, -1].contiguous() for w in srcs))
            store[id(blk)] = hit
        return hit[1]

    def chain_supported(self, cond):
        return cond.dim() == 3 and cond.shape[1] == 18 and cond.shape[2] == 512 and self.att_mapper[0].dim == 512

    def prepare_chain(self, cond, steps):
        """Everything that depends on (condition, step) only: e-parts of the four Linear(513) layers and the gamma/beta
        heads of every block for steps 0..steps-1 (two [steps*B*18, 512] GEMMs per block)."""
        B = cond.shape[0]
        M = B * 18
        state = []
        for blk in self.att_mapper:
            e = blk.embed(cond)
            heads = []
            for seq, key in ((blk.gamma_, "g"), (blk.beta_, "b")):
                pre = H.tacc_head_pre(e[key].reshape(M, 512), seq[0].weight[:, -1], seq[1].weight, seq[1].bias, steps,
                                      self.max_period)
                heads.append(H.linear(pre, seq[3].weight, seq[3].bias, act=seq.last).view(steps, B, 18, 512))
            wq, wk = self._tcols(blk)
            state.append({"eQ": e["Q"].reshape(M, 512), "ek": e["k"].reshape(M, 512), "gamma": heads[0], "beta": heads[1],
                          "wq": wq, "wk": wk, "wcat": self._wcat(blk), "wcat_frag": self._wcat_frag(blk)})
        return state

    def chain_step(self, x, pn, state, i, c1=None, c2=None, coef_idx=None):
        """One denoiser evaluation at step i on (x, pixelnorm(x)); with c1/c2 the update c1[k]*f(x) + c2[k]*x (k = coef_idx,
        default i: DDPM posterior mean; DDIM passes its own tables) is fused into the last block's tail.  Returns
        (new x or x0, its pixelnorm)."""
        B = x.shape[0]
        tf = float(i) / float(self.max_period)
        cur, cur_pn = x, pn
        last = len(self.att_mapper) - 1
        for bi, st in enumerate(state):
            P = H.gemm_nt(cur_pn.view(B * 18, 512), st["wcat"])
            t = H.tacc_chan_attn(P, st["ek"], st["wk"], tf, B)
            mix = bi == last and c1 is not None
            cur, cur_pn = H.tacc_tail(P, st["eQ"], st["wq"], tf, t, st["gamma"][i], st["beta"][i], B, xold=x if mix else None,
                                      c1=c1, c2=c2, idx=i if coef_idx is None else coef_idx)
        return cur, cur_pn


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """float64 schedule, reference ldm/util2.py:21-43."""
    if schedule == "linear":
        betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2
    elif schedule == "cosine":
        ts = torch.arange(n_timestep + 1, dtype=torch.float64) / n_timestep + cosine_s
        alphas = torch.cos(ts / (1 + cosine_s) * np.pi / 2).pow(2)
        alphas = alphas / alphas[0]
        betas = (1 - alphas[1:] / alphas[:-1]).clamp(0, 0.999)
    elif schedule == "sqrt_linear":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64)
    elif schedule == "sqrt":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64) ** 0.5
    else:
        raise ValueError(f"schedule '{schedule}' unknown.")
    return betas.numpy()


class My_DDPM(nn.Module):
    """x0-parameterised DDPM over W+ latents; inference = deterministic posterior-mean chain (reference
    ldm/ddpm.py:400-429: p_sample returns the mean, the drawn noise is unused)."""

    def __init__(self, denoise, timesteps=1000, beta_schedule="linear", clip_denoised=False, linear_start=1e-4,
                 linear_end=2e-2, cosine_s=8e-3, given_betas=None, v_posterior=0., l_simple_weight=1., parameterization="x0"):
        super().__init__()
        assert parameterization in ["eps", "x0"], 'currently only supporting "eps" and "x0"'
        self.parameterization = parameterization
        self.clip_denoised = clip_denoised
        self.model = denoise
        self.v_posterior, self.l_simple_weight = v_posterior, l_simple_weight
        self.register_schedule(given_betas, beta_schedule, timesteps, linear_start, linear_end, cosine_s)

    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        betas = given_betas if given_betas is not None else make_beta_schedule(beta_schedule, timesteps, linear_start,
                                                                               linear_end, cosine_s)
        alphas = 1. - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1., ac[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        f32 = partial(torch.tensor, dtype=torch.float32)
        pv = (1 - self.v_posterior) * betas * (1. - ac_prev) / (1. - ac) + self.v_posterior * betas
        for name, val in (("betas", betas), ("alphas_cumprod", ac), ("alphas_cumprod_prev", ac_prev),
                          ("sqrt_alphas_cumprod", np.sqrt(ac)), ("sqrt_one_minus_alphas_cumprod", np.sqrt(1. - ac)),
                          ("log_one_minus_alphas_cumprod", np.log(1. - ac)), ("sqrt_recip_alphas_cumprod", np.sqrt(1. / ac)),
                          ("sqrt_recipm1_alphas_cumprod", np.sqrt(1. / ac - 1)), ("posterior_variance", pv),
                          ("posterior_log_variance_clipped", np.log(np.maximum(pv, 1e-20))),
                          ("posterior_mean_coef1", betas * np.sqrt(ac_prev) / (1. - ac)),
                          ("posterior_mean_coef2", (1. - ac_prev) * np.sqrt(alphas) / (1. - ac))):
            self.register_buffer(name, f32(val))

    # ---- reference-shaped single step (used by callers that drive the loop themselves)
    @torch.no_grad()
    def p_sample(self, x, t, c, clip_denoised=True, repeat_noise=False):
        model_out = self.model(x, c, t)
        return self._posterior_mean(model_out, x, int(t[0])), model_out

    def _posterior_mean(self, model_out, x, i):
        if self.parameterization == "eps":
            x0 = self.sqrt_recip_alphas_cumprod[i] * x - self.sqrt_recipm1_alphas_cumprod[i] * model_out
        else:
            x0 = model_out
        if self.clip_denoised:
            x0 = x0.clamp(-1., 1.)
        return H.axpby_idx(x0.contiguous(), x.contiguous(), self.posterior_mean_coef1, self.posterior_mean_coef2, i)

    @torch.no_grad()
    def forward(self, x=None, condi_in=None, training=False, x_T=None, x_T_owned=False):
        """training=False: sample from x_T ~ N(0, I) (or the given `x_T`, an extension for parity runs) conditioned on
        `condi_in`; returns the last denoised latent.  x_T_owned: the caller hands over a scratch tensor (the pipeline's keyed
        draw) that the chain may update in place -- no protective copy."""
        if training:
            raise RuntimeError("My_DDPM.forward is the fused inference chain; the training-mode forward of code_diffuser_train.py is "
                               "vspbfr_amd.training.ddpm_training_forward(ddpm, x, condi_in)")
        cond = condi_in.contiguous()
        B, n_tok = cond.shape[0], cond.shape[1]
        x = x_T.contiguous() if x_T is not None else torch.randn(cond.shape, device=cond.device)
        if (self.parameterization == "x0" and not self.clip_denoised and hasattr(self.model, "chain_supported")
                and self.model.chain_supported(cond) and self.num_timesteps <= self.model.max_period):
            state = self.model.prepare_chain(cond, self.num_timesteps)
            x = x.clone() if (x_T is not None and not x_T_owned) else x  # the chain updates x in place
            return H.tacc_chain(x, state, list(reversed(range(self.num_timesteps))), c1=self.posterior_mean_coef1,
                                c2=self.posterior_mean_coef2, t_div=self.model.max_period)
        emb = self.model.embed(cond)  # step-independent half of every Linear(513)
        for i in reversed(range(self.num_timesteps)):
            t = torch.full((B,), i, device=cond.device, dtype=torch.long)
            x0 = self.model.run(x, self.model.condition(emb, t, n_tok))
            x = self._posterior_mean(x0, x, i)
        return x
