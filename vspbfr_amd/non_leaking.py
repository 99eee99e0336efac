"""Adaptive discriminator augmentation (ADA) of the training step over the gfx950 operators -- API mirror of the reference's
non_leaking.py (`AdaptiveAugment` :481-517, `augment` :930-934, `random_apply_affine` :857-907, `random_apply_color` :921-927;
SURVEY 8f row 4).

    augment(img, p, transform_matrix=(G, C)) -> (augmented, (G, C))

geometric part: reflect-pad by what the transformed corners need, 2x up-sampling with the 12-tap sym6 wavelet filter (separable
`op.upfirdn2d`, generic kernel), ONE fused affine-grid + bilinear sampling launch (`vsp_affine_sample_f32`: the reference
materialises the grid and calls aten::grid_sampler_2d), 2x down-sampling with the flipped filter; colour part: a per-image 3x4
matrix on the pixels (`vsp_color_affine_f32`).  Everything is differentiable in the image to second order (each op's adjoint is
another launch of the same family), which is what the generator step and the R1 penalty need when augmentation is on.
The transformation matrices are drawn on the host exactly as the ADA paper prescribes (flip, 90-degree rotations, integer and
fractional translations, isotropic / anisotropic scaling, arbitrary rotations; brightness, contrast, luma flip, hue rotation,
saturation -- each applied with probability p); passing `transform_matrix` pins them (tests, tests/golden/ada.npz)."""
import math

import torch
import torch.nn.functional as F
from torch.autograd import Function

from . import hip_ops
from ._lib import check, lib
from .op import upfirdn2d

SYM6 = (0.015404109327027373, 0.0034907120842174702, -0.11799011114819057, -0.048311742585633, 0.4910559419267466,
        0.787641141030194, 0.3379294217276218, -0.07263752278646252, -0.021060292512300564, 0.04472490177066578,
        0.0017677118642428036, -0.007800708325034148)


# ------------------------------------------------------------------------------------------------ device primitives
@hip_ops.device_guarded
def _affine_sample_raw(x, theta, out_hw):
    x = hip_ops._req(x.contiguous(), "x")
    theta = hip_ops._req(theta.contiguous(), "theta")
    B, C, IH, IW = x.shape
    out = torch.empty((B, C, out_hw[0], out_hw[1]), device=x.device, dtype=torch.float32)
    check(lib.vsp_affine_sample_f32(hip_ops._ptr(out), hip_ops._ptr(x), hip_ops._ptr(theta), B, C, IH, IW, out_hw[0], out_hw[1],
                                    hip_ops._stream()), "affine_sample")
    return out


@hip_ops.device_guarded
def _affine_sample_adjoint_raw(g, theta, in_hw):
    g = hip_ops._req(g.contiguous(), "g")
    theta = hip_ops._req(theta.contiguous(), "theta")
    B, C, OH, OW = g.shape
    gx = torch.empty((B, C, in_hw[0], in_hw[1]), device=g.device, dtype=torch.float32)
    check(lib.vsp_affine_sample_bwd_f32(hip_ops._ptr(gx), hip_ops._ptr(g), hip_ops._ptr(theta), B, C, in_hw[0], in_hw[1], OH, OW,
                                        hip_ops._stream()), "affine_sample_bwd")
    return gx


class _AffineSample(Function):
    """x -> grid_sample(x, affine_grid(theta)); linear in x, so forward and adjoint are each other's backward."""

    @staticmethod
    def forward(ctx, x, theta, out_hw):
        ctx.save_for_backward(theta)
        ctx.hw = (tuple(x.shape[2:]), tuple(out_hw))
        return _affine_sample_raw(x, theta, out_hw)

    @staticmethod
    def backward(ctx, g):
        (theta,) = ctx.saved_tensors
        return _AffineSampleAdjoint.apply(g, theta, ctx.hw[0]), None, None


class _AffineSampleAdjoint(Function):
    @staticmethod
    def forward(ctx, g, theta, in_hw):
        ctx.save_for_backward(theta)
        ctx.out_hw = tuple(g.shape[2:])
        return _affine_sample_adjoint_raw(g, theta, in_hw)

    @staticmethod
    def backward(ctx, gg):
        (theta,) = ctx.saved_tensors
        return _AffineSample.apply(gg, theta, ctx.out_hw), None, None


def affine_sample(x, theta, out_hw):
    """F.grid_sample(x, F.affine_grid(theta, (B, C, *out_hw), align_corners=False), "bilinear", "zeros", align_corners=False)."""
    if torch.is_grad_enabled() and x.requires_grad:
        return _AffineSample.apply(x, theta, tuple(out_hw))
    return _affine_sample_raw(x, theta, out_hw)


@hip_ops.device_guarded
def _color_raw(x, M, t):
    x = hip_ops._req(x.contiguous(), "x")
    B, C, Hh, Ww = x.shape
    if C != 3:
        raise RuntimeError("apply_color: 3-channel images")
    y = torch.empty_like(x)
    check(lib.vsp_color_affine_f32(hip_ops._ptr(y), hip_ops._ptr(x), hip_ops._ptr(hip_ops._req(M.contiguous(), "M")),
                                   hip_ops._ptr(hip_ops._req(t.contiguous(), "t")) if t is not None else None, B, Hh * Ww,
                                   hip_ops._stream()), "color_affine")
    return y


class _Color(Function):
    @staticmethod
    def forward(ctx, x, M, t):
        ctx.save_for_backward(M)
        return _color_raw(x, M, t)

    @staticmethod
    def backward(ctx, g):
        (M,) = ctx.saved_tensors
        return _Color.apply(g, M.transpose(1, 2).contiguous(), None), None, None


def apply_color(img, mat):
    """img (B,3,H,W), mat (B,4,4): pixel <- mat[:3,:3] pixel + mat[:3,3] (reference non_leaking.py:910-918)."""
    mat = mat.to(device=img.device, dtype=torch.float32)
    M, t = mat[:, :3, :3].contiguous(), mat[:, :3, 3].contiguous()
    if torch.is_grad_enabled() and img.requires_grad:
        return _Color.apply(img, M, t)
    return _color_raw(img, M, t)


# ------------------------------------------------------------------------------------------------ matrix sampling (host)
def _eye(n, size):
    return torch.eye(n).unsqueeze(0).repeat(size, 1, 1)


def _maybe(p, transform, prev):
    """apply `transform` to each sample with probability p"""
    size = transform.shape[0]
    pick = torch.empty(size).bernoulli_(p).view(size, 1, 1)
    return (pick * transform + (1 - pick) * _eye(transform.shape[1], size)) @ prev


def _mat2(entries, size):
    """(size, 3, 3) homogeneous 2-D matrices from a dict {(row, col): tensor (size,)}"""
    m = _eye(3, size)
    for (r, c), v in entries.items():
        m[:, r, c] = v
    return m


def sample_affine(p, size, height, width):
    """The geometric transformations of ADA (Karras et al. 2020, Sec. 2.2 / App. B), each with probability p (arbitrary
    rotations: 1 - sqrt(1 - p) before and after the anisotropic scaling); returns the FORWARD matrices (size, 3, 3)."""
    G = _eye(3, size)
    flip = torch.randint(0, 2, (size,)).float()
    G = _maybe(p, _mat2({(0, 0): 1 - 2 * flip}, size), G)                                         # x-flip
    # the reference draws the quarter-turn count from the two categories (0, 3) (non_leaking.py:673 `category_sample(size, (0, 3))`), not
    # from {0, 1, 2, 3}: one randint(high=2) index into that list
    th = -math.pi / 2 * torch.tensor((0.0, 3.0))[torch.randint(high=2, size=(size,))]
    G = _maybe(p, _mat2({(0, 0): th.cos(), (0, 1): -th.sin(), (1, 0): th.sin(), (1, 1): th.cos()}, size), G)   # 90-degree rotations
    t = torch.empty(2, size).uniform_(-0.125, 0.125)
    G = _maybe(p, _mat2({(0, 2): torch.round(t[1] * width), (1, 2): torch.round(t[0] * height)}, size), G)     # integer translation
    s = torch.empty(size).log_normal_(mean=0, std=0.2 * math.log(2))
    G = _maybe(p, _mat2({(0, 0): s, (1, 1): s}, size), G)                                         # isotropic scaling
    p_rot = 1 - math.sqrt(1 - p)

    def rot():
        a = -torch.empty(size).uniform_(-math.pi, math.pi)
        return _mat2({(0, 0): a.cos(), (0, 1): -a.sin(), (1, 0): a.sin(), (1, 1): a.cos()}, size)
    G = _maybe(p_rot, rot(), G)                                                                   # pre-rotation
    s = torch.empty(size).log_normal_(mean=0, std=0.2 * math.log(2))
    G = _maybe(p, _mat2({(0, 0): s, (1, 1): 1 / s}, size), G)                                     # anisotropic scaling
    G = _maybe(p_rot, rot(), G)                                                                   # post-rotation
    t = torch.empty(2, size).normal_(0, 0.125)
    return _maybe(p, _mat2({(0, 2): t[1] * width, (1, 2): t[0] * height}, size), G)               # fractional translation


def sample_color(p, size):
    """The colour transformations of ADA as (size, 4, 4) matrices on (r, g, b, 1)."""
    C = _eye(4, size)
    v = torch.full((3,), 1 / math.sqrt(3))                     # luma axis
    vv = torch.zeros(4, 4)
    vv[:3, :3] = torch.outer(v, v)
    b = torch.empty(size).normal_(0, 0.2)
    m = _eye(4, size)
    m[:, :3, 3] = b.view(-1, 1)
    C = _maybe(p, m, C)                                                                           # brightness
    c = torch.empty(size).log_normal_(mean=0, std=0.5 * math.log(2))
    m = _eye(4, size)
    m[:, 0, 0] = m[:, 1, 1] = m[:, 2, 2] = c
    C = _maybe(p, m, C)                                                                           # contrast
    i = torch.randint(0, 2, (size,)).float().view(-1, 1, 1)
    C = _maybe(p, _eye(4, size) - 2 * vv.unsqueeze(0) * i, C)                                     # luma flip
    a = torch.empty(size).uniform_(-math.pi, math.pi).view(-1, 1, 1)
    cross = torch.tensor([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])
    rot = a.cos() * torch.eye(3) + a.sin() * cross + (1 - a.cos()) * torch.outer(v, v)
    m = _eye(4, size)
    m[:, :3, :3] = rot
    C = _maybe(p, m, C)                                                                           # hue rotation
    s = torch.empty(size).log_normal_(mean=0, std=math.log(2)).view(-1, 1, 1)
    C = _maybe(p, vv.unsqueeze(0) + (_eye(4, size) - vv.unsqueeze(0)) * s, C)                     # saturation
    return C


# ------------------------------------------------------------------------------------------------ the augmentation
def _S(sx, sy):
    return torch.tensor(((sx, 0, 0), (0, sy, 0), (0, 0, 1)), dtype=torch.float32)


def _T(tx, ty):
    return torch.tensor(((1, 0, tx), (0, 1, ty), (0, 0, 1)), dtype=torch.float32)


def get_padding(G, height, width, kernel_size):
    """How far the inverse-transformed image corners reach outside the image (+ the filter's support): reflect padding per side,
    capped at size - 1 (reference non_leaking.py:770-790)."""
    cx, cy = (width - 1) / 2, (height - 1) / 2
    corners = torch.tensor([(-cx, -cy, 1), (cx, -cy, 1), (cx, cy, 1), (-cx, cy, 1)])
    cp = (G.cpu().float() @ corners.T)[:, :2, :].permute(1, 0, 2).flatten(1)     # (2, batch * 4): x row, y row
    reach = torch.cat((-cp, cp)).max(1).values                                   # (-x, -y, +x, +y)
    pad_k = kernel_size // 4
    reach = reach + torch.tensor([pad_k * 2 - cx, pad_k * 2 - cy] * 2)
    reach = reach.clamp(min=0).minimum(torch.tensor([width - 1, height - 1] * 2, dtype=torch.float32))
    px1, py1, px2, py2 = (int(v) for v in reach.ceil())
    return px1, px2, py1, py2


def random_apply_affine(img, p, G=None, antialiasing_kernel=SYM6):
    """`G` = the INVERSE transformation matrices (batch, 3, 3) in pixel units about the image centre (drawn when None), as in the
    reference (non_leaking.py:857-907)."""
    batch, channel, height, width = img.shape
    len_k = len(antialiasing_kernel)
    kernel = torch.as_tensor(antialiasing_kernel, dtype=torch.float32, device=img.device)
    kernel_flip = torch.flip(kernel, (0,))
    if G is None:
        G = torch.inverse(sample_affine(p, batch, height, width))
    G = G.cpu().float()
    px1, px2, py1, py2 = get_padding(G, height, width, len_k)
    img_pad = F.pad(img, (px1, px2, py1, py2), mode="reflect")
    G_inv = _T((px1 - px2) / 2, (py1 - py2) / 2) @ G
    up = ((len_k + 2 - 1) // 2, (len_k - 2) // 2)
    img_2x = upfirdn2d(img_pad, kernel.unsqueeze(0), up=(2, 1), pad=(up[0], up[1], 0, 0))
    img_2x = upfirdn2d(img_2x, kernel.unsqueeze(1), up=(1, 2), pad=(0, 0, up[0], up[1]))
    G_inv = _S(2, 2) @ G_inv @ _S(1 / 2, 1 / 2)
    G_inv = _T(-0.5, -0.5) @ G_inv @ _T(0.5, 0.5)
    pad_k = len_k // 4
    oh, ow = (height + pad_k * 2) * 2, (width + pad_k * 2) * 2
    G_inv = _S(2 / img_2x.shape[3], 2 / img_2x.shape[2]) @ G_inv @ _S(1 / (2 / ow), 1 / (2 / oh))
    img_affine = affine_sample(img_2x, G_inv[:, :2, :].to(img.device), (oh, ow))
    d_p = -pad_k * 2
    down = (d_p + (len_k - 2 + 1) // 2, d_p + (len_k - 2) // 2)
    out = upfirdn2d(img_affine, kernel_flip.unsqueeze(0), down=(2, 1), pad=(down[0], down[1], 0, 0))
    out = upfirdn2d(out, kernel_flip.unsqueeze(1), down=(1, 2), pad=(0, 0, down[0], down[1]))
    return out, G


def random_apply_color(img, p, C=None):
    if C is None:
        C = sample_color(p, img.shape[0])
    return apply_color(img, C), C


def augment(img, p, transform_matrix=(None, None)):
    img, G = random_apply_affine(img, p, transform_matrix[0])
    img, C = random_apply_color(img, p, transform_matrix[1])
    return img, (G, C)


class AdaptiveAugment:
    """The probability controller (reference non_leaking.py:481-517): every `update_every` calls the sign statistic r_t =
    E[sign(D(real))] summed over the ranks decides whether p moves up or down by n / ada_aug_len."""

    def __init__(self, ada_aug_target, ada_aug_len, update_every, device):
        self.ada_aug_target, self.ada_aug_len, self.update_every = ada_aug_target, ada_aug_len, update_every
        self.ada_update = 0
        self.ada_aug_buf = torch.tensor([0.0, 0.0], device=device)
        self.r_t_stat = 0
        self.ada_aug_p = 0

    @torch.no_grad()
    def tune(self, real_pred):
        self.ada_aug_buf += torch.tensor((torch.sign(real_pred).sum().item(), real_pred.shape[0]), device=real_pred.device)
        self.ada_update += 1
        if self.ada_update % self.update_every == 0:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                dist.all_reduce(self.ada_aug_buf, op=dist.ReduceOp.SUM)
            pred_signs, n_pred = self.ada_aug_buf.tolist()
            self.r_t_stat = pred_signs / n_pred
            sign = 1 if self.r_t_stat > self.ada_aug_target else -1
            self.ada_aug_p = min(1, max(0, self.ada_aug_p + sign * n_pred / self.ada_aug_len))
            self.ada_aug_buf.mul_(0)
            self.ada_update = 0
        return self.ada_aug_p
