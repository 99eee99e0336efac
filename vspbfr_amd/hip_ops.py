"""Tensor-level wrappers over the C ABI (include/vspbfr_hip.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; all arithmetic happens in
libvspbfr_hip.so.  Every wrapper requires CUDA(HIP) fp32 contiguous tensors and raises RuntimeError otherwise --
there is no CPU path (the reference's own CPU branch lives in oracle/, test-only).
"""
import ctypes as C
import math

import os

import torch

from . import _lib
from ._lib import ConvParams, FirEpilogue, GemmParams, check, lib, tune_env

SQRT2 = math.sqrt(2.0)


class ConvProfiler:
    """Optional per-launch timing of vsp_conv2d_f32 with HIP events on the launch stream (bench.py's roofline leg):
    records (algorithmic FLOPs, start event, end event) for every conv launch while installed as `hip_ops.PROFILER`."""

    def __init__(self):
        self.records = []

    def begin(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def end(self, start, flops, tag, nbytes=0):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        self.records.append((flops, start, ev, tag, nbytes))

    def summary(self):
        """-> (total algorithmic FLOPs, total ms, launches); call after a device synchronize."""
        fl = sum(r[0] for r in self.records)
        ms = sum(r[1].elapsed_time(r[2]) for r in self.records)
        return fl, ms, len(self.records)

    def executed_flops(self):
        """FLOPs the matrix pipe actually EXECUTED: a Winograd F(2x2,3x3) launch runs 16 multiplies per 2x2 output tile where the
        direct form (the algorithmic count of `summary`) has 36; every other kernel executes its algorithmic count."""
        f = {"wino": 16.0 / 36.0, "wino4": 36.0 / 144.0, "wino4f": 36.0 / 144.0}   # F(4x4,3x3): 36 multiplies per 4x4 tile against 144
        return sum(r[0] * (f.get(r[3][7], 1.0) if len(r[3]) > 7 else 1.0) for r in self.records)

    def by_kind(self):
        """{kernel family: (algorithmic FLOPs, ms, launches)} -- direct / pipelined / tconv / wino / bf16."""
        out = {}
        for r in self.records:
            k = r[3][7] if len(r[3]) > 7 else "?"
            a = out.setdefault(k, [0.0, 0.0, 0])
            a[0] += r[0]
            a[1] += r[1].elapsed_time(r[2])
            a[2] += 1
        return {k: (v[0], v[1], v[2]) for k, v in out.items()}

    def algorithmic_bytes(self):
        """Sum over the launches of the bytes a convolution must move when every operand crosses HBM exactly once: input image,
        output, weights, noise map and residuals at their element sizes (the unit of the HBM roofline)."""
        return sum(r[4] for r in self.records)


WGRAD_WORKSPACE = True   # False: vsp_conv2d_wgrad_f32 reduces its split-K partial sums with atomics
PROFILER = None
RECORDER = None  # tools/autotune_conv.py: list collecting the shape key of every conv launch


def _load_tune_table():
    import json
    import os
    path = os.environ.get("VSPBFR_CONV_TUNE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "conv_tune.json")  # (override: tuning runs)
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f)


def _config_ids():
    return {lib.vsp_conv2d_config_name(i).decode(): i + 1 for i in range(lib.vsp_conv2d_num_configs())}


# shape key -> 1-based tile configuration id measured fastest on MI355X (tools/autotune_conv.py stores configuration NAMES,
# resolved here); shapes that are not in the table use the library's cost model (tile_hint = 0)
CONFIG_IDS = _config_ids()
TUNE = {k: CONFIG_IDS[v] for k, v in _load_tune_table().items() if v in CONFIG_IDS}
# shapes where a Winograd kernel won: True = F(2x2,3x3) (vsp_conv2d_winograd_f32), 4 = F(4x4,3x3) (vsp_conv2d_winograd4_f32, deep layers)
#   5 = F(4x4,3x3) fused in registers (vsp_conv2d_winograd4f_f32, shallow wide layers)
#   (also the SMART dilation-group launches of 128 / 256 channels: one launch, a partition of workgroups per group, dilated groups through the
#   kernel's LDS window loader)
WINO = {k: ({"winograd4": 4, "winograd4f": 5}.get(v, True)) for k, v in _load_tune_table().items()
        if v in ("winograd", "winograd4", "winograd4f")}


def _load_bf16_tune(name="conv_tune_bf16.json"):
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), name)
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return {k: int(v) for k, v in json.load(f).items()}


# shape key -> tile variant of vsp_conv2d_bf16 measured fastest INSIDE the pipeline (tools/autotune_bf16.py); other shapes use the
# library's rule.  BF16_FORCE (tuner only): variant tried on every launch it fits.
BF16_TUNE = _load_bf16_tune()
BF16X3_TUNE = _load_bf16_tune("conv_tune_bf16x3.json")
BF16_FORCE = 0


# the data gradient of the four dilated SMART branches as one convolution (conv_pipe.hip MODE 3) instead of a grouped conv + a sum over
# the branches (tools / A-B runs can switch it off)
SMART_ADJOINT_ONE_PASS = tune_env("VSP_SMART_ADJOINT_ONE_PASS", "1") != "0"
# training: demodulation coefficients and their gradient on the fused kernels (vsp_demod_weight_f32) instead of torch autograd
FUSED_DEMOD_GRAD = tune_env("VSP_FUSED_DEMOD_GRAD", "1") != "0"
# training: a discriminator ResBlock of the first-order passes as one autograd node (discriminator._ResBlockFO)
RESBLOCK_ONE_NODE = tune_env("VSP_RESBLOCK_ONE_NODE", "1") != "0"


def conv_key(B, Cin, H, W, pc, OH, OW):
    return f"{B},{Cin},{H},{W},{pc.G},{pc.cout_g},{pc.kh},{pc.kw},{pc.stride},{pc.dil[0]},{OH},{OW}" + (
        f",g{pc.x_group_stride}" if pc.x_group_stride else "") + (",q" if pc.dil_by_input_quarter else "")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def operand_device(args, kwargs=None):
    """Device index shared by every tensor among an operator's arguments (tensors inside lists / tuples included), None when there is
    no tensor; RuntimeError when they lie on different devices.  Pure host logic (CPU-testable): the C ABI takes raw pointers and one
    stream, so operands on two devices cannot be served by one launch -- the reference's extensions would fault on the foreign pointer,
    this refuses."""
    dev = None
    stack = list(args) + (list(kwargs.values()) if kwargs else [])
    while stack:
        a = stack.pop()
        if isinstance(a, torch.Tensor):
            if dev is None:
                dev = a.device
            elif a.device != dev:
                raise RuntimeError(f"operands on different devices: {dev} and {a.device}")
        elif isinstance(a, (list, tuple)):
            stack.extend(a)
    return dev


def device_guarded(fn):
    """The reference's `at::cuda::OptionalCUDAGuard device_guard(device_of(input))` (op/fused_bias_act.cpp:25, op/upfirdn2d.cpp:23) for the
    ctypes path: the launch goes to the TENSORS' device and that device's current stream (`_stream()` is evaluated inside the guard), not
    to whatever device is current in the calling thread; the previous device comes back on return.  One process per GPU never takes the
    slow branch."""
    import functools

    @functools.wraps(fn)
    def guarded(*args, **kwargs):
        dev = operand_device(args, kwargs)
        if dev is None or dev.type != "cuda" or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)

    guarded.__wrapped_op__ = fn
    return guarded


def _req(t, name, bf16_ok=False):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f"{name} must be a tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")
    if t.dtype != torch.float32 and not (bf16_ok and t.dtype == torch.bfloat16):
        raise RuntimeError(f"{name} must be float32{' or bfloat16' if bf16_ok else ''} (got {t.dtype})")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return t


# bf16 ACTIVATIONS in HBM (BASELINE configs[2]; needs BF16_CONV = True).  Off: every tensor is fp32.  On: a layer that runs on a
# bf16-I/O kernel (vsp_conv2d_bf16 with io_bf16, vsp_upfirdn2d_bf16, vsp_pointwise_bf16) reads and writes bf16 tensors; the dtype
# simply follows the kernel, and a consumer whose kernel wants the other type converts its (small-map) input first -- the
# transitions sit at the 16^2 maps, where the bf16 conv kernel hands over to the fp32 ones.
ACT_BF16 = False
BF = torch.bfloat16


def to_bf16(t):
    """fp32 -> bf16 (round to nearest even) through vsp_convert_f32_to_bf16; bf16 tensors pass through."""
    if t is None or t.dtype == BF:
        return t
    t = _req(t, "tensor")
    out = torch.empty(t.shape, device=t.device, dtype=BF)
    check(lib.vsp_convert_f32_to_bf16(_ptr(out), _ptr(t), t.numel(), _stream()), "convert_f32_to_bf16")
    return out


def to_f32(t):
    """bf16 -> fp32 (exact) through vsp_convert_bf16_to_f32; fp32 tensors pass through."""
    if t is None or t.dtype == torch.float32:
        return t
    t = _req(t, "tensor", bf16_ok=True)
    out = torch.empty(t.shape, device=t.device, dtype=torch.float32)
    check(lib.vsp_convert_bf16_to_f32(_ptr(out), _ptr(t), t.numel(), _stream()), "convert_bf16_to_f32")
    return out


def as_dtype(t, dtype):
    return to_bf16(t) if dtype == BF else to_f32(t)


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _opt(t, name):
    return None if t is None else _req(t, name)


# ----------------------------------------------------------------------------------------------- fused_bias_act
def fused_bias_act(input, bias, refer, act, grad, alpha, scale):
    """Same signature as the reference's native `fused.fused_bias_act` (op/fused_bias_act.cpp:18-31):
    empty `bias` / `refer` tensors mean "absent"; returns a new tensor."""
    x = _req(input, "input")
    b = _req(bias, "bias") if bias is not None and bias.numel() else None
    r = _req(refer, "refer") if refer is not None and refer.numel() else None
    out = torch.empty_like(x)
    step_b = 1
    for i in range(2, x.dim()):
        step_b *= x.size(i)
    size_b = b.numel() if b is not None else 0
    if r is not None and r.numel() != x.numel():
        raise RuntimeError("refer must have the same number of elements as input")
    check(lib.vsp_fused_bias_act_f32(_ptr(out), _ptr(x), _ptr(b), _ptr(r), x.numel(), step_b, size_b, int(act),
                                     int(grad), float(alpha), float(scale), _stream()), "fused_bias_act")
    return out


# ----------------------------------------------------------------------------------------------- upfirdn2d
def fir_out_size(in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, px0, px1, py0, py1):
    return ((in_h * up_y + py0 + py1 - kh + down_y) // down_y, (in_w * up_x + px0 + px1 - kw + down_x) // down_x)


def upfirdn2d_native_layout(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1, epilogue=None):
    """Same contract as the reference's native `upfirdn2d_op.upfirdn2d` (op/upfirdn2d.cpp:17-31):
    input [major, in_h, in_w, minor] -> new tensor [major, out_h, out_w, minor]."""
    x = _req(input, "input", bf16_ok=True)
    k = _req(kernel, "kernel")
    if x.dim() != 4 or k.dim() != 2:
        raise RuntimeError("upfirdn2d expects input [major,H,W,minor] and a 2-D kernel")
    major, in_h, in_w, minor = x.shape
    kh, kw = k.shape
    out_h, out_w = fir_out_size(in_h, in_w, kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)
    if out_h < 0 or out_w < 0:
        raise RuntimeError(f"upfirdn2d: negative output size {out_h}x{out_w}")
    out = torch.empty((major, out_h, out_w, minor), device=x.device, dtype=x.dtype)
    epi = C.byref(epilogue) if epilogue is not None else None
    fn = lib.vsp_upfirdn2d_bf16 if x.dtype == BF else lib.vsp_upfirdn2d_f32
    check(fn(_ptr(out), _ptr(x), _ptr(k), major, in_h, in_w, minor, kh, kw, up_x, up_y, down_x,
             down_y, pad_x0, pad_x1, pad_y0, pad_y1, epi, _stream()), "upfirdn2d")
    return out


def fir_bf16_ok(kh, kw, up, down, minor, out_w):
    """What vsp_upfirdn2d_bf16 serves: the blur form."""
    return up == 1 and down == 1 and minor == 1 and out_w >= 16 and (kh, kw) in ((4, 4), (3, 3), (2, 2))


SEPARABLE_BLUR = tune_env("VSP_FIR_SEPARABLE", "1") != "0"   # (A/B runs: the 2-D form)


def taps_separable(kernel):
    """True when the 2-D taps are exactly an outer product k[i][j] == k[i][0] * k[0][j] / k[0][0] in fp32 (make_kernel of a 1-D list: every
    blur of the path) -- decided once per tap tensor on the host (one small device-to-host copy at first use); the answer rides on the
    tensor OBJECT (a module buffer lives as long as its module; a temporary takes its answer with it -- an address is not an identity)."""
    hit = getattr(kernel, "_vsp_separable", None)
    if hit is not None and hit[0] == kernel._version:
        return hit[1]
    k = kernel.detach().float().cpu()
    sep = False
    if k.dim() == 2:
        k = k.flip(0, 1)        # the kernels work on the flipped taps (a true convolution): row factor = first column / corner
        sep = bool(float(k[0, 0]) != 0.0 and torch.equal(k, ((k[:, :1] / k[0, 0]) * k[:1, :]).float()))
    try:
        kernel._vsp_separable = (kernel._version, sep)
    except (AttributeError, RuntimeError):
        pass
    return sep


def blur_fused(x, kernel, pad, plane_scale=None, noise=None, noise_w=None, act_bias=None, act=False, res1=None,
               res2=None, slope=0.2, gain=SQRT2):
    """NCHW blur (up=down=1) with the fused demod/noise/bias/leaky-relu/residual epilogue of the C ABI.  The element type of
    x decides the kernel (fp32 / bf16 activations); residuals are converted to it when they differ (small maps only).  Outer-product
    taps are flagged (VSP_FIR_SEPARABLE): the kernels may then run a row pass and a column pass."""
    x = _req(x, "x", bf16_ok=True)
    B, Cc, H, W = x.shape
    kh, kw = kernel.shape
    if x.dtype == BF and not fir_bf16_ok(kh, kw, 1, 1, 1, fir_out_size(H, W, kh, kw, 1, 1, 1, 1, pad[0], pad[1], pad[0], pad[1])[1]):
        x = to_f32(x)
    res1, res2 = as_dtype(res1, x.dtype), as_dtype(res2, x.dtype)
    epi = FirEpilogue()
    keep = [_opt(plane_scale, "plane_scale"), _opt(noise, "noise"), _opt(noise_w, "noise_w"), _opt(act_bias, "act_bias"),
            _req(res1, "res1", True) if res1 is not None else None, _req(res2, "res2", True) if res2 is not None else None]
    epi.plane_scale, epi.noise, epi.noise_w, epi.act_bias, epi.res1, epi.res2 = [
        (t.data_ptr() if t is not None else None) for t in keep]
    epi.channels, epi.act, epi.slope, epi.gain = Cc, 1 if act else 0, slope, gain
    epi.flags = _lib.FIR_SEPARABLE if (SEPARABLE_BLUR and taps_separable(kernel)) else 0
    out = upfirdn2d_native_layout(x.view(B * Cc, H, W, 1), kernel, 1, 1, 1, 1, pad[0], pad[1], pad[0], pad[1], epi)
    return out.view(B, Cc, out.shape[1], out.shape[2])


# ----------------------------------------------------------------------------------------------- conv2d
class PackedConv:
    """A convolution weight in the kernel's layout Wp[g][tap][ci][co_g] (include/vspbfr_hip.h) plus its geometry.
    Built once per device on first use (pack_weight below; cached on the owning module)."""

    __slots__ = ("w", "G", "cout_g", "cin", "kh", "kw", "stride", "dil", "pad_y", "pad_x", "x_group_stride", "dil_by_input_quarter",
                 "_wino", "_wino4", "_wino4f", "_bf16", "_bf16x3", "_bf16rv", "_bf16dg")

    def __init__(self, w, G, cout_g, cin, kh, kw, stride=1, dil=(1,), pad_y=(0,), pad_x=None, x_group_stride=0, dil_by_input_quarter=False):
        self.w, self.G, self.cout_g, self.cin, self.kh, self.kw = w, G, cout_g, cin, kh, kw
        self.stride = stride
        # True: the data gradient of four dilated branches in one pass (include/vspbfr_hip.h: dil_by_input_quarter) -- G = 4 blocks of
        # cout_g output channels over ALL cin input channels, dil / pad belong to the input-channel quarter
        self.dil_by_input_quarter = bool(dil_by_input_quarter)
        self.x_group_stride = x_group_stride  # > 0: true grouped conv, group g reads input channels [g*stride, +cin)
        rep = lambda t, fill: tuple(t) * 4 if len(t) == 1 else tuple(t) + (fill,) * (4 - len(t))  # noqa: E731
        self.dil = rep(dil, 1)          # one value = the same geometry for every group
        self.pad_y = rep(pad_y, 0)
        self.pad_x = rep(pad_y if pad_x is None else pad_x, 0)
        self._wino = None
        self._wino4 = None
        self._wino4f = None
        self._bf16 = None
        self._bf16x3 = None
        self._bf16rv = None
        self._bf16dg = None

    @property
    def cout(self):
        return self.G * self.cout_g

    def winograd_weight(self):
        """U = G g G^T (16, Cin, Cout), built on first use and kept with the packed weight."""
        if self._wino is None:
            with torch.no_grad():
                self._wino = winograd_weight(self.w)
        return self._wino


    def winograd4_weight(self):
        """U = G g G^T of F(4x4,3x3) (36, Cin, Cout) in the fragment order of vsp_conv2d_winograd4_f32, built on first use."""
        if self._wino4 is None:
            with torch.no_grad():
                self._wino4 = winograd4_weight(self.w)
        return self._wino4

    def winograd4f_weight(self):
        """U = G g G^T of F(4x4,3x3) in the order of the fused kernel vsp_conv2d_winograd4f_f32, built on first use."""
        if self._wino4f is None:
            with torch.no_grad():
                self._wino4f = winograd4f_weight(self.w)
        return self._wino4f

    def bf16_weight(self):
        """The weight rounded to bf16 in the LDS-image order of vsp_conv2d_bf16, built on first use."""
        if self._bf16 is None:
            with torch.no_grad():
                self._bf16 = bf16_weight(self.w)
        return self._bf16


    def bf16rv_weight(self):
        """The weight rounded to bf16 in the fragment order of vsp_conv2d_bf16rv, built on first use."""
        if self._bf16rv is None:
            with torch.no_grad():
                self._bf16rv = bf16rv_weight(self.w)
        return self._bf16rv

    def bf16dg_weight(self):
        """The fp32 weight in the A-fragment order of vsp_conv2d_bf16dg, built on first use."""
        if self._bf16dg is None:
            with torch.no_grad():
                self._bf16dg = bf16dg_weight(self.w)
        return self._bf16dg

    def bf16x3_weight(self):
        """hi + lo bf16 parts of the weight in the LDS-image order of vsp_conv2d_bf16x3, built on first use."""
        if self._bf16x3 is None:
            with torch.no_grad():
                self._bf16x3 = bf16x3_weight(self.w)
        return self._bf16x3


# "bf16 kernels" configuration (BASELINE configs[2]): eligible convolutions run on vsp_conv2d_bf16 (bf16 MFMA, fp32
# accumulate, fp32 activations in HBM).  Off by default: the parity path is fp32 end to end.
BF16_CONV = False   # True: bf16 operands; "x3": split precision (hi + lo bf16 pairs, three MFMAs per product: fp32-grade results)
_env_dtype = os.environ.get("VSPBFR_CONV_DTYPE", "")   # process-wide default of the switch: "bf16" / "bf16x3" (entry points set it explicitly)
if _env_dtype:
    if _env_dtype not in ("f32", "bf16", "bf16x3"):
        raise RuntimeError(f"VSPBFR_CONV_DTYPE must be f32, bf16 or bf16x3 (got {_env_dtype!r})")
    BF16_CONV = {"f32": False, "bf16": True, "bf16x3": "x3"}[_env_dtype]


def bf16_eligible(pc, H, W, OH, OW, transposed=False, out_stride=(1, 1), out_offset=(0, 0)):
    """What vsp_conv2d_bf16 serves: 3x3 kernels with >= 16 input channels as (a) stride 1, padding = dilation, G <= 4 dilation
    groups over one input or true groups; (b) stride 2, dilation 1, padding 0 or 1, G = 1 or true groups; (c) the stride-2
    transposed conv (G = 1); dense output for (a) and (b)."""
    if pc.kh != 3 or pc.kw != 3 or pc.cin < 16 or pc.cin % 8 or pc.dil_by_input_quarter:
        return False
    if transposed:
        return pc.G == 1
    if tuple(out_stride) != (1, 1) or tuple(out_offset) != (0, 0):
        return False
    if pc.stride == 2:
        return (pc.dil[0] == 1 and pc.pad_y[0] == pc.pad_x[0] and pc.pad_y[0] in (0, 1) and (pc.G == 1 or pc.x_group_stride > 0)
                and (OH, OW) == conv2d_out_size(H, W, pc))
    if pc.stride != 1 or (pc.G > 4 and pc.x_group_stride == 0):
        return False
    if any(pc.pad_y[g] != pc.dil[g] or pc.pad_x[g] != pc.dil[g] for g in range(min(pc.G, 4))):
        return False
    return (OH, OW) == (H, W)


def bf16_profitable(pc, H, W, OH, OW, transposed=False):
    """Shapes on which the bf16 kernel beats the fp32 kernels (tools/bench_bf16.py, tools/conv_breakdown.py): 32-pixel row
    segments need maps (polyphase sub-images for dilation groups) at least 8-16 pixels wide."""
    if transposed:
        return W >= 16
    if pc.stride == 2:
        return OW >= 8
    dmax = max(pc.dil[:min(pc.G, 4)])
    return W >= 16 and H // dmax >= 2


def bf16_weight(wp):
    """packed weights (G, 9, Cin, cout_g) fp32 -> bf16 in the LDS-image order of vsp_conv2d_bf16:
    [group][chunk][tap][octet 2][co_pad][8] with ci = 16 chunk + 8 octet + j; Cin zero-padded to 16, cout_g to 32."""
    ng, T, cin, cout = wp.shape
    nch, co_pad = (cin + 15) // 16, (cout + 31) // 32 * 32
    Wz = wp.new_zeros(ng, T, nch * 16, co_pad)
    Wz[:, :, :cin, :cout] = wp
    Wz = Wz.view(ng, T, nch, 2, 8, co_pad).permute(0, 2, 1, 3, 5, 4)
    return Wz.to(torch.bfloat16).contiguous().view(-1)


def bf16rv_weight(wp):
    """packed weights (G, 9, Cin, cout_g) fp32 -> per group [chunk Cin/8][ky 3][quad 2][half 2][co_pad][8] bf16 with element e = channel
    8 chunk + 4 quad + 2 half + (e >> 2), horizontal tap kx = e & 3 (kx = 3: zero), co_pad = cout_g rounded up to 32 (zero rows) --
    the A fragments of vsp_conv2d_bf16rv (G > 1: the dilation groups of one launch, one after the other)."""
    ng, T, cin, cout = wp.shape
    if T != 9 or cin % 8 or cout % 8:
        raise RuntimeError("bf16rv_weight: 3x3, Cin % 8 == 0, channels per group % 8 == 0")
    co_pad = (cout + 31) // 32 * 32
    Wz = wp.new_zeros(ng, 3, 4, cin, co_pad)
    Wz[:, :, :3, :, :cout] = wp.view(ng, 3, 3, cin, cout)
    Wz = Wz.view(ng, 3, 4, cin // 8, 2, 2, 2, co_pad).permute(0, 3, 1, 4, 5, 7, 6, 2)   # group, chunk, ky, quad, half, co, pair, kx
    return Wz.to(torch.bfloat16).contiguous().view(-1)


def bf16rv_eligible(pc, H, W, OH, OW, transposed=False, out_stride=(1, 1), out_offset=(0, 0)):
    """What vsp_conv2d_bf16rv serves (bf16 activations on top; the entry re-checks alignment and returns VSP_ENOTSUP): plain 3x3 layers
    and the 2-4 dilation groups of a SMART branch launch (dilations from 1, 2, 4, 8 over one input)."""
    if (transposed or pc.kh != 3 or pc.kw != 3 or pc.stride != 1 or pc.cin % 8 or pc.cin > 256 or W % 64 or (OH, OW) != (H, W)
            or tuple(out_stride) != (1, 1) or tuple(out_offset) != (0, 0) or pc.dil_by_input_quarter):
        return False
    if pc.G == 1:
        return pc.dil[0] == 1 and pc.pad_y[0] == 1 and pc.pad_x[0] == 1 and pc.cout_g % 32 == 0
    return (2 <= pc.G <= 4 and pc.x_group_stride == 0 and pc.cout_g % 8 == 0
            and all(pc.dil[g] in (1, 2, 4, 8) and pc.pad_y[g] == pc.dil[g] and pc.pad_x[g] == pc.dil[g] for g in range(pc.G)))


def bf16dg_weight(wp):
    """packed weights (G, 9, Cin, cout_g) fp32 -> fp32 [group][stage Cin/32][tap 9][lane 64][8] with lane = 16 kb + co, element e = input
    channel 32 stage + 8 kb + e (zero-padded: Cin to 32 stages, cout_g to 16) -- the A fragments of vsp_conv2d_bf16dg, which multiplies them
    by the image's style scale and rounds to bf16 itself."""
    ng, T, cin, cout = wp.shape
    if T != 9 or cout > 16 or cin > 64:
        raise RuntimeError("bf16dg_weight: 3x3, at most 16 channels per group, at most 64 input channels")
    nst = 1 if cin <= 32 else 2
    Wz = wp.new_zeros(ng, 9, nst * 32, 16)
    Wz[:, :, :cin, :cout] = wp
    return Wz.view(ng, 9, nst, 4, 8, 16).permute(0, 2, 1, 3, 5, 4).contiguous().view(-1)


def bf16dg_eligible(pc, H, W, OH, OW, transposed=False, out_stride=(1, 1), out_offset=(0, 0), in_shift=None):
    """What vsp_conv2d_bf16dg serves (bf16 activations on top): the 2-4 dilation groups of a SMART branch launch with at most 16 channels
    per group over at most 64 input channels."""
    if (transposed or pc.kh != 3 or pc.kw != 3 or pc.stride != 1 or pc.cin % 8 or pc.cin > 64 or pc.cout_g > 16 or W % 8 or (OH, OW) != (H, W)
            or tuple(out_stride) != (1, 1) or tuple(out_offset) != (0, 0) or pc.dil_by_input_quarter or pc.x_group_stride or in_shift is not None):
        return False
    return 1 <= pc.G <= 4 and all(pc.dil[g] in (1, 2, 4, 8) and pc.pad_y[g] == pc.dil[g] and pc.pad_x[g] == pc.dil[g] for g in range(pc.G))


BF16_MODW = tune_env("VSP_BF16_MODW", "1") != "0"   # per-image modulated weights (vsp_modulate_weight_bf16) on the modulated layers of vsp_conv2d_bf16


def bf16_modulated_weight(pc, in_scale):
    """(B sets of bf16(W * style[b]) in the LDS-image order of vsp_conv2d_bf16, byte stride between them): the reference's own fused
    modulated convolution (models/RestoreNet.py:381-383) -- one small launch per layer and batch; the conv kernel then copies its pixels."""
    in_scale = _req(in_scale, "in_scale")
    B = in_scale.shape[0]
    nbytes = lib.vsp_modulate_weight_bf16_bytes(pc.G, pc.cin, pc.cout_g)
    out = torch.empty((B, nbytes // 2), device=in_scale.device, dtype=BF)
    check(lib.vsp_modulate_weight_bf16(_ptr(out), _ptr(pc.w), _ptr(in_scale), B, in_scale.stride(0), pc.G, pc.cin, pc.cout_g, _stream()),
          "modulate_weight")
    return out, nbytes


BF16_DG = tune_env("VSP_BF16_DG", "1") != "0"   # the dilation-group kernel on the launches it serves (G > 1)
BF16_RV = tune_env("VSP_BF16_RV", "1") != "0"   # the row-vector-K kernel on the layers bf16rv_profitable names


def bf16rv_profitable(pc, H, W):
    """Layers where the row-vector-K kernel beats vsp_conv2d_bf16 (tools/bench_bf16rv.py, batch 16: 64 -> 64 at 512^2 x1.53, at 256^2
    x1.51, 64 -> 128 at 128^2 x1.46, 128 -> 128 at 256^2 x1.27, 32 -> 32 at 1024^2 x1.23, 256 -> 256 at 128^2 x1.10; 128 -> 128 at 64^2 x0.83)."""
    if pc.G > 1:   # dilation groups (served, tested, not chosen): every group stages the whole input patch for its few channels -- alone
        return False   # 64 -> 4 x 16 at 512^2 x1.12, 128 -> 4 x 32 at 256^2 x1.03, 256 -> 4 x 64 at 128^2 x0.83; inside configs[2] no gain (400-415 img/s)
    return pc.cin <= 256 and H * W >= 128 * 128


def _bf16rv_call(p, pc, keep, forced):
    """One launch of vsp_conv2d_bf16rv on the filled parameter block; False = the entry does not serve it (alignment): the caller
    goes on to vsp_conv2d_bf16."""
    bw = pc.bf16rv_weight()
    w0, h0 = p.w, p.tile_hint
    p.w, p.tile_hint = bw.data_ptr(), (p.tile_hint if forced else 0)
    rc = lib.vsp_conv2d_bf16rv(C.byref(p), _stream())
    if rc == -3 and not forced:
        p.w, p.tile_hint = w0, h0
        return False
    check(rc, "conv2d_bf16rv")
    keep.append(bw)
    return True


def _bf16dg_call(p, pc, keep, forced):
    """One launch of vsp_conv2d_bf16dg; False = the entry does not serve it (alignment): the caller goes on to the other bf16 kernels."""
    bw = pc.bf16dg_weight()
    w0, h0 = p.w, p.tile_hint
    p.w, p.tile_hint = bw.data_ptr(), 0
    rc = lib.vsp_conv2d_bf16dg(C.byref(p), _stream())
    if rc == -3 and not forced:
        p.w, p.tile_hint = w0, h0
        return False
    check(rc, "conv2d_bf16dg")
    keep.append(bw)
    return True


def bf16x3_weight(wp):
    """packed weights (G, 9, Cin, cout_g) fp32 -> [group][chunk][part 2][tap][octet 2][co_pad][8] bf16 with part 0 = bf16(W),
    part 1 = bf16(W - float(bf16(W))) (vsp_conv2d_bf16x3)."""
    ng, T, cin, cout = wp.shape
    nch, co_pad = (cin + 15) // 16, (cout + 31) // 32 * 32
    Wz = wp.new_zeros(ng, T, nch * 16, co_pad)
    Wz[:, :, :cin, :cout] = wp
    hi = Wz.to(torch.bfloat16)
    lo = (Wz - hi.float()).to(torch.bfloat16)
    parts = torch.stack([hi, lo], 0).view(2, ng, T, nch, 2, 8, co_pad).permute(1, 3, 0, 2, 4, 6, 5)
    return parts.contiguous().view(-1)


def pack_weight(weight, groups=1, adjoint=False, flip=False, scale=1.0, out=None):
    """(Cout, Cin, KH, KW) -> [G][KH*KW][Cin][Cout/G] contiguous, times `scale`, taps reversed when `flip`.  `adjoint` (one group):
    the weight of the data gradient, channels exchanged: [KH*KW][i = Cout][o = Cin] (= packing weight.transpose(0, 1)).
    Device weights are packed by vsp_pack_weight_f32 (one launch); the torch expression serves host tensors (layout tests)."""
    cout, cin, kh, kw = weight.shape
    assert cout % groups == 0 and (groups == 1 or not adjoint)
    if weight.is_cuda:
        w = _req(weight.detach().contiguous(), "weight")
        shape = (1, kh * kw, cout, cin) if adjoint else (groups, kh * kw, cin, cout // groups)
        wp = torch.empty(shape, device=w.device, dtype=torch.float32) if out is None else _req(out, "out")   # out: a slice of a group stack
        assert tuple(wp.shape) == shape
        check(lib.vsp_pack_weight_f32(_ptr(wp), _ptr(w), groups, cout // groups, cin, kh, kw, int(adjoint), int(flip), float(scale),
                                      _stream()), "pack_weight")
        return wp
    w = weight * scale if scale != 1.0 else weight
    if flip:
        w = w.flip(2, 3)
    if adjoint:
        w = w.transpose(0, 1)
        cout, cin = cin, cout
    wp = w.reshape(groups, cout // groups, cin, kh * kw).permute(0, 3, 2, 1).contiguous()
    return wp if out is None else out.copy_(wp)


def pack_weight_stack(weights, adjoint=False, flip=False, scale=1.0):
    """One packed weight with a group per entry of `weights` (equal shapes): the dilated branches of one layer."""
    cout, cin, kh, kw = weights[0].shape
    shape = (len(weights), kh * kw, cout, cin) if adjoint else (len(weights), kh * kw, cin, cout)
    wp = torch.empty(shape, device=weights[0].device, dtype=torch.float32)
    for i, w in enumerate(weights):
        pack_weight(w, 1, adjoint, flip, scale, out=wp[i:i + 1])
    return wp


def winograd_eligible(pc, H, W, OH, OW, transposed=False, out_stride=(1, 1), out_offset=(0, 0)):
    """3x3, stride 1, padding = dilation, G = 1 or up to four dilation groups over one shared input, dense output."""
    if transposed or pc.kh != 3 or pc.kw != 3 or pc.stride != 1 or pc.x_group_stride != 0 or not 1 <= pc.G <= 4 or pc.dil_by_input_quarter:
        return False
    if any(pc.pad_y[g] != pc.dil[g] or pc.pad_x[g] != pc.dil[g] or pc.dil[g] not in (1, 2, 4, 8) for g in range(pc.G)):
        return False
    return (OH, OW) == (H, W) and tuple(out_stride) == (1, 1) and tuple(out_offset) == (0, 0)


def winograd_weight(wp):
    """packed weights (G, 9, Cin, cout_g) -> U = G g G^T in the FRAGMENT order of vsp_conv2d_winograd_f32:
    [group][co tile][chunk][wave 8][lane 64][pp 2][mb MB] with position = 2 wave + pp, ci = CK chunk + (lane >> 4),
    co = 16 MB tile + 16 mb + (lane & 15); Cin / cout_g zero-padded to multiples of CK / 16 MB.  Sums in float64, rounded once
    (vsp_winograd_weight_f32: one launch -- a trained weight is transformed again every iteration)."""
    wp = _req(wp, "packed weight")
    ng, cin, cout = wp.shape[0], wp.shape[2], wp.shape[3]
    U = torch.empty(lib.vsp_winograd_weight_floats(ng, cin, cout), device=wp.device, dtype=torch.float32)
    check(lib.vsp_winograd_weight_f32(_ptr(U), _ptr(wp), ng, cin, cout, _stream()), "winograd_weight")
    return U


def winograd4_weight(wp):
    """packed weights (1, 9, Cin, Cout) -> U = G g G^T of F(4x4,3x3) (points 0, +-3/4, +-3/2, inf; fp64 sums rounded once) in the
    fragment order of vsp_conv2d_winograd4_f32 (vsp_winograd4_weight_f32)."""
    wp = _req(wp, "packed weight")
    ng, taps, cin, cout = wp.shape
    if ng != 1 or taps != 9:
        raise RuntimeError("winograd4_weight: one group of 3x3 taps")
    U = torch.empty(lib.vsp_winograd4_weight_floats(cin, cout), device=wp.device, dtype=torch.float32)
    check(lib.vsp_winograd4_weight_f32(_ptr(U), _ptr(wp), cin, cout, _stream()), "winograd4_weight")
    return U


def winograd4f_weight(wp):
    """packed weights (G, 9, Cin, cout_g) -> U = G g G^T of F(4x4,3x3) in the order of vsp_conv2d_winograd4f_f32
    ([group][co / 32][ci / 4][position pair][lane][4], include/vspbfr_hip.h)."""
    wp = _req(wp, "packed weight")
    ng, taps, cin, cout = wp.shape
    if taps != 9:
        raise RuntimeError("winograd4f_weight: 3x3 taps")
    n = lib.vsp_winograd4f_weight_floats(cin, cout)
    U = torch.empty(ng * n, device=wp.device, dtype=torch.float32)
    for g in range(ng):
        check(lib.vsp_winograd4f_weight_f32(_ptr(U[g * n:]), _ptr(wp[g]), cin, cout, _stream()), "winograd4f_weight")
    return U


def winograd4f_eligible(pc, H, W, OH, OW, transposed=False, out_stride=(1, 1), out_offset=(0, 0), in_shift=None):
    """fused F(4x4,3x3): one group or up to four dilation groups over one shared input, 3x3 / stride 1 / padding = dilation in (1, 2, 4, 8)
    -- dilated groups run on their polyphase sub-images: whole 4x4 tiles in each (H, W multiples of 4 x dilation) --, no affine shift, Cin a
    multiple of 8 up to 512, rows of at least 16 pixels"""
    return (winograd_eligible(pc, H, W, OH, OW, transposed, out_stride, out_offset) and in_shift is None
            and pc.cin % 8 == 0 and pc.cin <= 512 and W >= 16 and all(H % (4 * pc.dil[g]) == 0 and W % (4 * pc.dil[g]) == 0 for g in range(pc.G)))


def winograd4_eligible(pc, H, W, OH, OW, transposed=False, out_stride=(1, 1), out_offset=(0, 0), in_shift=None):
    """F(4x4,3x3) pair of kernels: one group, 3x3 / stride 1 / dilation 1 / padding 1, no affine shift, whole 4x4 tiles"""
    return (winograd_eligible(pc, H, W, OH, OW, transposed, out_stride, out_offset) and pc.G == 1 and pc.dil[0] == 1 and in_shift is None
            and pc.cin % 4 == 0 and H % 4 == 0 and W % 4 == 0)


def conv2d_out_size(H, W, pc):
    d, py, px = pc.dil[0], pc.pad_y[0], pc.pad_x[0]
    oh = (H + 2 * py - d * (pc.kh - 1) - 1) // pc.stride + 1
    ow = (W + 2 * px - d * (pc.kw - 1) - 1) // pc.stride + 1
    return oh, ow


def conv2d_packed(x, pc, out=None, out_hw=None, y_coff=0, out_stride=(1, 1), out_offset=(0, 0), in_scale=None,
                  in_scale_per_sample=True, in_shift=None, out_scale=None, ch_scale=None, ch_bias=None, act1=False,
                  bias1=None, noise=None, noise_w=None, act2=0, bias2=None, prelu=None, slope2=0.2, gain2=SQRT2, res1=None,
                  res2=None, res_coff=0, n_out=None, tile_hint=0, transposed=False, winograd=None, bf16=None, wino_form=0):
    """Launch vsp_conv2d_f32.  `out` (B, y_ch, y_h, y_w) is allocated when None.  `n_out` = (OH, OW) positions to
    compute (defaults to the standard conv output size).  `winograd`: True / False forces / forbids the F(2x2,3x3) kernel
    (vsp_conv2d_winograd_f32) on an eligible layer; None = what the tuned table says for this shape; `wino_form` names the F(2x2) kernel
    form (0 automatic, 1 task list, 2 row owner, 3 register-resident U: include/vspbfr_hip.h, tests / tuning).  `bf16`: True runs the
    layer on vsp_conv2d_bf16 (tile_hint = its variant), None = the module switch BF16_CONV on eligible layers."""
    x = _req(x, "x", bf16_ok=True)
    B, x_ch, H, W = x.shape
    Cin = pc.cin
    if (pc.G - 1) * pc.x_group_stride + Cin != x_ch:
        raise RuntimeError(f"conv2d: input has {x_ch} channels, weight expects {(pc.G - 1) * pc.x_group_stride + Cin}")
    OH, OW = n_out if n_out is not None else ((H + 1, W + 1) if transposed else conv2d_out_size(H, W, pc))
    # ---- which kernel family serves this launch (decided before anything is allocated: it fixes the activation dtype)
    key = conv_key(B, Cin, H, W, pc, OH, OW) + (",t" if transposed else "") + (",s" if in_shift is not None else "")
    if tile_hint == 0 and TUNE:
        pref = TUNE.get(key, 0)
        if pref == 0 and B != 8:  # the table was measured at batch 8; large layers keep their tile at other batches
            pref = TUNE.get("8" + key[key.index(","):], 0)
        tile_hint = -pref  # negative = preference: falls back to the cost model when it cannot serve this call's operands
    bf_ok = bf16_eligible(pc, H, W, OH, OW, transposed, out_stride, out_offset)
    x3 = bf16 == "x3" or (bf16 is None and BF16_CONV == "x3")
    rv = None          # the row-vector-K kernel (vsp_conv2d_bf16rv): "rv" forces it, None = where it is eligible and measured faster
    dg = None          # the dilation-group kernel (vsp_conv2d_bf16dg): "dg" forces it, None = the dilation-group launches it serves
    if bf16 == "rv":
        bf16, rv = True, True
    if bf16 == "dg":
        bf16, dg = True, True
    if bf16 is None:
        bf16 = bool(BF16_CONV) and bf_ok and not winograd and bf16_profitable(pc, H, W, OH, OW, transposed)
        if x3 and bf16:  # split precision: doubled LDS images -- the layers where it beats the tuned fp32 kernels (tools/conv_breakdown.py)
            bf16 = (transposed and W >= 32) or (not transposed and (
                pc.stride == 1 or (pc.stride == 2 and pc.G == 1 and OW >= 32)))
    elif bf16 and not bf_ok:
        raise RuntimeError("conv2d: this layer is not eligible for the bf16 kernel (see bf16_eligible)")
    if rv and not bf16rv_eligible(pc, H, W, OH, OW, transposed, out_stride, out_offset):
        raise RuntimeError("conv2d: this layer is not eligible for the row-vector bf16 kernel (see bf16rv_eligible)")
    if dg and not bf16dg_eligible(pc, H, W, OH, OW, transposed, out_stride, out_offset, in_shift):
        raise RuntimeError("conv2d: this layer is not eligible for the dilation-group bf16 kernel (see bf16dg_eligible)")
    rv = rv or dg   # (the same activation-type requirement below)
    if rv and not (x.dtype == BF or (ACT_BF16 and out is None)) or (rv and out is not None and out.dtype != BF):
        raise RuntimeError("conv2d: bf16='rv' names the row-vector kernel, which reads and writes bf16 activations: pass a bf16 input "
                           "(and output) or switch ACT_BF16 on")
    # bf16 activations: the bf16 kernel (not its split-precision form) reads and writes bf16 when the configuration asks for it or
    # the caller hands it a bf16 tensor; every other kernel is fp32 on both sides
    io_bf = bool(bf16) and not x3 and (x.dtype == BF or (ACT_BF16 and out is None)) and (out is None or out.dtype == BF)
    act_dt = BF if io_bf else torch.float32
    x = as_dtype(x, act_dt)
    res1, res2 = as_dtype(res1, act_dt), as_dtype(res2, act_dt)
    if out is None and transposed:
        out = torch.empty((B, pc.cout, 2 * H + 1, 2 * W + 1), device=x.device, dtype=act_dt)
    if out is None:
        yh, yw = out_hw if out_hw is not None else (OH, OW)
        out = torch.empty((B, pc.cout, yh, yw), device=x.device, dtype=act_dt)
    _req(out, "out", bf16_ok=io_bf)
    if out.dtype != act_dt:
        raise RuntimeError(f"conv2d: `out` is {out.dtype} but this launch writes {act_dt}")
    if B == 0:  # empty batch: nothing to enqueue (an empty tensor has no device pointer to hand to the C ABI)
        return out
    p = ConvParams()
    keep = [x, pc.w, out, _opt(in_scale, "in_scale"), _opt(in_shift, "in_shift"), _opt(out_scale, "out_scale"),
            _opt(ch_scale, "ch_scale"), _opt(ch_bias, "ch_bias"), _opt(bias1, "bias1"), _opt(noise, "noise"),
            _opt(noise_w, "noise_w"), _opt(bias2, "bias2"), _opt(prelu, "prelu"),
            _req(res1, "res1", io_bf) if res1 is not None else None, _req(res2, "res2", io_bf) if res2 is not None else None]
    dp = [(t.data_ptr() if t is not None else None) for t in keep]
    (p.x, p.w, p.y, p.in_scale, p.in_shift, p.out_scale, p.ch_scale, p.ch_bias, p.bias1, p.noise, p.noise_w, p.bias2,
     p.prelu, p.res1, p.res2) = dp
    p.io_bf16 = 1 if io_bf else 0
    p.B, p.Cin, p.H, p.W = B, Cin, H, W
    p.G, p.cout_g, p.OH, p.OW, p.KH, p.KW = pc.G, pc.cout_g, OH, OW, pc.kh, pc.kw
    p.stride_y = p.stride_x = pc.stride
    for g in range(4):
        p.dil[g], p.pad_y[g], p.pad_x[g] = pc.dil[g], pc.pad_y[g], pc.pad_x[g]
    p.y_ch, p.y_coff, p.y_h, p.y_w = out.shape[1], y_coff, out.shape[2], out.shape[3]
    p.osy, p.osx = out_stride
    p.ooy, p.oox = out_offset
    p.in_scale_bstride = (in_scale.shape[-1] if pc.x_group_stride else Cin) if (in_scale is not None and in_scale_per_sample) else 0
    p.act1, p.slope1, p.gain1 = (1 if act1 else 0), 0.2, SQRT2
    p.act2, p.slope2, p.gain2 = int(act2), float(slope2), float(gain2)
    rt = res1 if res1 is not None else res2
    p.res_ch = rt.shape[1] if rt is not None else 0
    p.res_coff = res_coff
    if bf16:
        winograd = False
        if tile_hint < 0:
            tile_hint = 0
    wino_ok = winograd_eligible(pc, H, W, OH, OW, transposed, out_stride, out_offset)
    named_fused = winograd == 5       # the caller asked for the fused F(4x4) kernel by name: no silent fall-back (tests, tuners)
    if winograd is None:
        winograd = wino_ok and tile_hint == 0 and WINO.get(key, WINO.get("8" + key[key.index(","):], False))
    elif winograd and not wino_ok:
        raise RuntimeError("conv2d: this layer is not eligible for the Winograd kernel (3x3, stride 1, dilation 1, pad 1, G = 1)")
    if winograd == 4 and winograd is not True and not winograd4_eligible(pc, H, W, OH, OW, transposed, out_stride, out_offset, in_shift):
        winograd = True     # (the deep-layer form does not serve this call's operands: F(2x2,3x3))
    if winograd == 5 and not winograd4f_eligible(pc, H, W, OH, OW, transposed, out_stride, out_offset, in_shift):
        if named_fused:
            raise RuntimeError("conv2d: this launch is not eligible for the fused F(4x4,3x3) kernel (winograd4f_eligible: one group or up to four "
                               "dilation groups over one shared input, 3x3 / stride 1 / padding = dilation in {1, 2, 4, 8}, H and W multiples of "
                               "4 x dilation, no affine shift, Cin % 8 == 0 up to 512, W >= 16)")
        winograd = True
    if RECORDER is not None:
        RECORDER.append((key, (B, Cin, H, W, OH, OW), pc, transposed))
    p.tile_hint = tile_hint
    p.x_ch, p.x_group_stride = x_ch, pc.x_group_stride
    p.transposed = 1 if transposed else 0
    p.dil_by_input_quarter = 1 if pc.dil_by_input_quarter else 0
    if rt is not None and (rt.shape[0] != B or rt.shape[2] != out.shape[2] or rt.shape[3] != out.shape[3]):
        raise RuntimeError("conv2d: residual must match the output tensor's batch and spatial size")
    prof = PROFILER
    if prof is not None:
        start = prof.begin()
    ran_rv = False
    if bf16 and x3:
        bw = pc.bf16x3_weight()
        keep.append(bw)
        p.w = bw.data_ptr()
        if tile_hint == 0:
            p.tile_hint = BF16X3_TUNE.get(key, BF16X3_TUNE.get("8" + key[key.index(","):], 0))
            if BF16_FORCE:
                p.tile_hint = BF16_FORCE
        rc = lib.vsp_conv2d_bf16x3(C.byref(p), _stream())
        if rc != 0 and BF16_FORCE and tile_hint == 0:  # tuner: the forced variant does not serve this launch
            p.tile_hint = 0
            rc = lib.vsp_conv2d_bf16x3(C.byref(p), _stream())
        if rc == -3:  # VSP_ENOTSUP: the doubled LDS images of this shape do not fit (small stride-2 maps) -> the fp32 kernel
            p.w, bf16 = pc.w.data_ptr(), False
            check(lib.vsp_conv2d_f32(C.byref(p), _stream()), "conv2d")
        else:
            check(rc, "conv2d_bf16x3")
    elif bf16 and io_bf and (dg or (dg is None and BF16_DG and pc.G > 1 and tile_hint == 0 and not BF16_FORCE and rv is None)) and bf16dg_eligible(
            pc, H, W, OH, OW, transposed, out_stride, out_offset, in_shift) and _bf16dg_call(p, pc, keep, dg):
        ran_rv = "dg"
    elif bf16 and not dg and (rv or (rv is None and BF16_RV and tile_hint == 0 and not BF16_FORCE)) and io_bf and bf16rv_eligible(
            pc, H, W, OH, OW, transposed, out_stride, out_offset) and (rv or bf16rv_profitable(pc, H, W)) and _bf16rv_call(p, pc, keep, rv):
        ran_rv = True
    elif bf16:
        modw = (BF16_MODW and io_bf and in_scale is not None and in_scale_per_sample and in_shift is None and pc.x_group_stride == 0
                and W % 2 == 0 and in_scale.dim() == 2 and in_scale.shape == (B, Cin) and x.data_ptr() % 4 == 0)
        if modw:
            # the style goes into per-image weights (the reference's fused form): the kernel's staging becomes a copy (conv_bf16.hip NOSC)
            bw, wbytes = bf16_modulated_weight(pc, in_scale)
            p.in_scale, p.in_scale_bstride, p.w_bstride = None, 0, wbytes
        else:
            bw = pc.bf16_weight()
        keep.append(bw)
        p.w = bw.data_ptr()

        def launch_bf16(strict=True):
            rc = lib.vsp_conv2d_bf16(C.byref(p), _stream())
            if rc == -3 and p.w_bstride:   # VSP_ENOTSUP: this tile's patch plane is too large for the copy-only staging -> shared weights + style scale
                b2 = pc.bf16_weight()
                keep.append(b2)
                p.w, p.w_bstride, p.in_scale, p.in_scale_bstride = b2.data_ptr(), 0, in_scale.data_ptr(), Cin
                rc = lib.vsp_conv2d_bf16(C.byref(p), _stream())
            if strict:
                check(rc, "conv2d_bf16")
            return rc
        if tile_hint == 0:
            p.tile_hint = BF16_TUNE.get(key, BF16_TUNE.get("8" + key[key.index(","):], 0))
            if BF16_FORCE:
                p.tile_hint = BF16_FORCE
                if launch_bf16(strict=False) != 0:  # the forced variant does not serve this launch
                    p.tile_hint = 0
                    launch_bf16()
            else:
                launch_bf16()
        else:
            launch_bf16()
    elif winograd:
        rc = -3
        if winograd == 5:
            u4 = pc.winograd4f_weight()
            keep.append(u4)
            p.w = u4.data_ptr()
            rc = lib.vsp_conv2d_winograd4f_f32(C.byref(p), _stream())
            if rc not in (0, -3) or (rc == -3 and named_fused):
                check(rc, "conv2d_winograd4f")
            if rc == -3:   # VSP_ENOTSUP: alignment of an operand plane -> F(2x2,3x3)
                winograd = True
        elif winograd == 4 and winograd is not True:
            u4 = pc.winograd4_weight()
            nfl = lib.vsp_conv2d_winograd4_work_floats(C.byref(p))
            work = torch.empty(nfl, device=x.device, dtype=torch.float32)   # V = B^T d B in fragment order (2.25 x the input)
            keep += [u4, work]
            p.w = u4.data_ptr()
            rc = lib.vsp_conv2d_winograd4_f32(C.byref(p), _ptr(work), nfl, _stream())
            if rc not in (0, -3):
                check(rc, "conv2d_winograd4")
            if rc == -3:   # VSP_ENOTSUP: alignment of an operand plane -> F(2x2,3x3)
                winograd = True
        if rc == -3:
            uw = pc.winograd_weight()
            keep.append(uw)
            p.w = uw.data_ptr()
            p.tile_hint = wino_form
            check(lib.vsp_conv2d_winograd_f32(C.byref(p), _stream()), "conv2d_winograd")
    else:
        check(lib.vsp_conv2d_f32(C.byref(p), _stream()), "conv2d")
    if prof is not None:
        es = 2 if io_bf else 4
        n_out_el = B * pc.cout * ((2 * H + 1) * (2 * W + 1) if transposed else OH * OW)
        nbytes = (x.numel() + n_out_el * (1 + (res1 is not None) + (res2 is not None))) * es + pc.cout * Cin * pc.kh * pc.kw * (2 if bf16 else 4) + (
            B * OH * OW * 4 if noise is not None else 0)
        prof.end(start, 2.0 * B * pc.cout * (H * W if transposed else OH * OW) * Cin * pc.kh * pc.kw, (Cin, pc.cout, OH, OW, pc.kh, pc.stride, pc.G, ("bf16x3" if x3 else (("bf16dg" if ran_rv == "dg" else "bf16rv") if ran_rv else "bf16")) if bf16 else (("wino4f" if winograd == 5 else ("wino4" if (winograd == 4 and winograd is not True) else "wino")) if winograd else ("tconv" if transposed else "direct")), key), nbytes)
    return out


CONV1X1_SMALL_MAX_P = 1024   # maps up to 32 x 32 (measured to 28 x 28, batch 8: tools/bench_1x1_gemm.py); larger maps amortise the tiled kernel's weight slabs


def conv1x1_small(x, w2d, bias=None, act=False, slope=0.2, gain=SQRT2):
    """y[b, co, p] = act(w2d @ x[b] + bias) for a 1x1 stride-1 conv on a small map (vsp_conv1x1_small_f32); w2d = (Cout, Cin)."""
    x, w2d = _req(x, "x"), _req(w2d, "weight")
    B, cin, Hh, Ww = x.shape
    cout = w2d.shape[0]
    if w2d.shape[1] != cin:
        raise RuntimeError(f"conv1x1_small: input has {cin} channels, weight expects {w2d.shape[1]}")
    y = torch.empty((B, cout, Hh, Ww), device=x.device, dtype=torch.float32)
    check(lib.vsp_conv1x1_small_f32(_ptr(y), _ptr(w2d), _ptr(x), _ptr(_opt(bias, "bias")), B, cout, cin, Hh * Ww, int(bool(act)),
                                    float(slope), float(gain), _stream()), "conv1x1_small")
    return y


def conv2d(x, weight, bias=None, stride=1, padding=0, dilation=1, **epi):
    """F.conv2d-shaped convenience entry (groups=1): packs the weight on the fly (tests, generic callers)."""
    weight = _req(weight, "weight")
    cout, cin, kh, kw = weight.shape
    if (kh == 1 and kw == 1 and stride == 1 and padding == 0 and cin % 16 == 0 and x.dtype == torch.float32 and x.dim() == 4
            and x.shape[2] * x.shape[3] <= CONV1X1_SMALL_MAX_P and not (set(epi) - {"act2", "bias2", "slope2", "gain2"})
            and not (bias is not None and epi.get("bias2") is not None)):
        act = epi.get("act2", 0)
        if act in (0, 1):   # GEMM-shaped: K = Cin against a few hundred columns
            return conv1x1_small(x, weight.view(cout, cin), bias if bias is not None else epi.get("bias2"), act == 1,
                                 epi.get("slope2", 0.2), epi.get("gain2", SQRT2))
    # The packed (and, on a Winograd layer, transformed) weight rides on the tensor object it was built from: a long-lived weight
    # (frozen loss networks, folded BatchNorm weights) is packed once per version, a temporary dies with its packing.
    key = (weight._version, weight.data_ptr(), stride, dilation, padding)
    cached = getattr(weight, "_vsp_packed", None)
    if cached is not None and cached[0] == key:
        pc = cached[1]
    else:
        pc = PackedConv(pack_weight(weight), 1, cout, cin, kh, kw, stride, (dilation,), (padding,))
        try:
            weight._vsp_packed = (key, pc)
        except (AttributeError, RuntimeError):
            pass
    return conv2d_packed(x, pc, ch_bias=bias, **epi)


# transposed conv, stride 2, pad 0, as four sub-pixel phases (include/vspbfr_hip.h, osy/osx/ooy/oox)
def pack_transposed_s2(weight_oihw):
    """weight (Cout, Cin, 3, 3) as stored by ModulatedConv2d (reference models/RestoreNet.py:463-465; the reference
    transposes it to (Cin, Cout, 3, 3) for conv_transpose2d, :527-529).  out[2m+py, 2n+px] only sees taps with
    ky = py (mod 2): phase 0 is a 2-tap correlation [W[2], W[0]] with pad 1 over m = 0..H, phase 1 the single tap W[1]
    over m = 0..H-1.  Returns {(py, px): PackedConv}."""
    cout, cin, kh, kw = weight_oihw.shape
    assert kh == 3 and kw == 3
    sel = {0: [2, 0], 1: [1]}
    phases = {}
    for py in (0, 1):
        for px in (0, 1):
            sub = weight_oihw[:, :, sel[py], :][:, :, :, sel[px]]
            phases[(py, px)] = PackedConv(pack_weight(sub), 1, cout, cin, len(sel[py]), len(sel[px]), 1, (1,),
                                          (1 if py == 0 else 0,), (1 if px == 0 else 0,))
    return phases


def conv_transpose2d_s2_fused(x, pc, **kw):
    """(B, Cin, H, W) -> (B, Cout, 2H+1, 2W+1): conv_transpose2d(stride=2, padding=0), 3x3, in ONE launch (`pc` is the
    ordinary packed 3x3 weight).  Supports in_scale / out_scale / per-channel epilogue terms."""
    return conv2d_packed(x, pc, transposed=True, **kw)


def conv_transpose2d_s2_into(x, pc, out_hw, **kw):
    """The same launch writing into a (B, Cout, OH, OW) tensor with OH in {2H+1, 2H+2} (the data gradient of a stride-2 conv over an
    even-sized input has one zero row / column more): the margin is zeroed here, no padded copy is made."""
    B, _, H, W = x.shape
    OH, OW = out_hw
    if not (0 <= OH - (2 * H + 1) <= 1 and 0 <= OW - (2 * W + 1) <= 1):
        raise RuntimeError("conv_transpose2d_s2_into: out_hw must be (2H+1 or 2H+2, 2W+1 or 2W+2)")
    out = torch.empty((B, pc.cout, OH, OW), device=x.device, dtype=torch.float32)
    if OH > 2 * H + 1:
        out[:, :, -1].zero_()
    if OW > 2 * W + 1:
        out[:, :, :, -1].zero_()
    return conv2d_packed(x, pc, transposed=True, out=out, bf16=False, **kw)


def conv_transpose2d_s2(x, phases, **kw):
    """(B, Cin, H, W) -> (B, Cout, 2H+1, 2W+1): conv_transpose2d(stride=2, padding=0) with a 3x3 kernel, as four
    sub-pixel phase launches (kept as the cross-check of the fused kernel)."""
    B, _, H, W = x.shape
    cout = phases[(0, 0)].cout
    out = torch.empty((B, cout, 2 * H + 1, 2 * W + 1), device=x.device, dtype=x.dtype)
    for (py, px), pc in phases.items():
        conv2d_packed(x, pc, out=out, out_stride=(2, 2), out_offset=(py, px),
                      n_out=(H + 1 if py == 0 else H, W + 1 if px == 0 else W), **kw)
    return out


# ----------------------------------------------------------------------------------------------- gemm / linear
def gemm_nt(a, b, out=None, alpha=1.0, bias=None, bias_scale=1.0, act=0, slope=0.2, gain=SQRT2, a_strides=None,
            b_strides=None, dims=None, bias_zs=0):
    """C[z,m,n] = epi(alpha * sum_k A[z,m,k] B[z,n,k]).  Default: a [Z?,M,K], b [Z?,N,K] contiguous.
    `a_strides` = (zs, ms, ks), `b_strides` = (zs, ns, ks) and dims = (Z, M, N, K) describe arbitrary views of the
    storage starting at a.data_ptr()/b.data_ptr()."""
    if dims is None:
        if a.dim() == 2:
            a3, b3 = a.unsqueeze(0), b.unsqueeze(0)
        else:
            a3, b3 = a, b
        Z, M, K = a3.shape
        N = b3.shape[1]
        if b3.shape[0] not in (1, Z) or b3.shape[2] != K:
            raise RuntimeError("gemm_nt: shape mismatch")
        a_strides = (a3.stride(0), a3.stride(1), a3.stride(2))
        b_strides = ((b3.stride(0) if b3.shape[0] == Z and Z > 1 else 0), b3.stride(1), b3.stride(2))
        if Z == 1:
            a_strides = (0,) + a_strides[1:]
        out_shape = (M, N) if a.dim() == 2 else (Z, M, N)
    else:
        Z, M, N, K = dims
        out_shape = (Z, M, N)
    for t, nm in ((a, "a"), (b, "b")):
        if not (t.is_cuda and t.dtype == torch.float32):
            raise RuntimeError(f"gemm_nt: {nm} must be a CUDA float32 tensor")
    if out is None:
        out = torch.empty(out_shape, device=a.device, dtype=a.dtype)
    _req(out, "out")
    p = GemmParams()
    p.A, p.Bm, p.C = a.data_ptr(), b.data_ptr(), out.data_ptr()
    p.Z, p.M, p.N, p.K = Z, M, N, K
    p.a_zs, p.a_ms, p.a_ks = a_strides
    p.b_zs, p.b_ns, p.b_ks = b_strides
    p.c_zs, p.c_ms = M * N, N
    p.alpha = alpha
    p.bias = _opt(bias, "bias").data_ptr() if bias is not None else None
    p.bias_scale, p.act, p.slope, p.gain = bias_scale, act, slope, gain
    p.bias_zs = bias_zs
    check(lib.vsp_gemm_f32(C.byref(p), _stream()), "gemm")
    return out


def linear(x, weight, bias=None, alpha=1.0, bias_scale=1.0, act=0):
    """F.linear(x, weight*alpha, bias*bias_scale) (+ fused leaky-relu*sqrt2 when act=1, sigmoid when act=2)."""
    lead = x.shape[:-1]
    if x.dim() == 2 and x.stride(1) == 1 and x.is_cuda and x.dtype == torch.float32:
        x2 = x  # rows may be strided (a slice latent[:, i] of a (B, 18, D) tensor): the GEMM takes the row pitch, no copy
    else:
        x2 = _req(x, "x").reshape(-1, x.shape[-1])
    out = gemm_nt(x2, _req(weight, "weight"), alpha=alpha, bias=bias, bias_scale=bias_scale, act=act)
    return out.view(*lead, weight.shape[0])


# ----------------------------------------------------------------------------------------------- row helpers
def pixelnorm_dim1(x, eps=1e-8):
    x = _req(x, "x")
    if x.dim() == 2:
        Z, R, Cc = 1, x.shape[0], x.shape[1]
        # PixelNorm on (B, D) normalises over dim 1 = D: that is a [B, D, 1] problem
        Z, R, Cc = x.shape[0], x.shape[1], 1
    else:
        Z, R = x.shape[0], x.shape[1]
        Cc = x.numel() // (Z * R) if Z * R else 0
    out = torch.empty_like(x)
    check(lib.vsp_pixelnorm_dim1_f32(_ptr(out), _ptr(x), Z, R, Cc, eps, _stream()), "pixelnorm_dim1")
    return out


def layernorm(x, add=None, gamma=None, beta=None, eps=1e-5, post_lrelu=False):
    x = _req(x, "x")
    cols = x.shape[-1]
    rows = x.numel() // cols
    out = torch.empty_like(x)
    check(lib.vsp_layernorm_f32(_ptr(out), _ptr(x), _ptr(_opt(add, "add")), _ptr(_opt(gamma, "gamma")),
                                _ptr(_opt(beta, "beta")), rows, cols, eps, 1 if post_lrelu else 0, 0.2, SQRT2,
                                _stream()), "layernorm")
    return out


def softmax_lastdim(x):
    x = _req(x, "x")
    cols = x.shape[-1]
    out = torch.empty_like(x)
    check(lib.vsp_softmax_lastdim_f32(_ptr(out), _ptr(x), x.numel() // cols, cols, _stream()), "softmax_lastdim")
    return out


def softmax_dim1(x):
    x = _req(x, "x")
    Z, R = x.shape[0], x.shape[1]
    Cc = x.numel() // (Z * R)
    out = torch.empty_like(x)
    check(lib.vsp_softmax_dim1_f32(_ptr(out), _ptr(x), Z, R, Cc, _stream()), "softmax_dim1")
    return out


def film(h, gamma, beta):
    out = torch.empty_like(_req(h, "h"))
    check(lib.vsp_film_f32(_ptr(out), _ptr(h), _ptr(_req(gamma, "gamma")), _ptr(_req(beta, "beta")), h.numel(),
                           _stream()), "film")
    return out


def axpby_idx(x, y, a, b, idx):
    out = torch.empty_like(_req(x, "x"))
    check(lib.vsp_axpby_idx_f32(_ptr(out), _ptr(x), _ptr(_req(y, "y")), _ptr(_req(a, "a")), _ptr(_req(b, "b")), int(idx),
                                x.numel(), _stream()), "axpby_idx")
    return out


def demod_coefs(style, wsq, wscale, eps=1e-8):
    style, wsq = _req(style, "style"), _req(wsq, "wsq")
    B, Cin = style.shape
    Cout = wsq.shape[0]
    out = torch.empty((B, Cout), device=style.device, dtype=style.dtype)
    check(lib.vsp_demod_f32(_ptr(out), _ptr(style), _ptr(wsq), B, Cin, Cout, wscale, eps, _stream()), "demod")
    return out


def demod_weight(style, weight, wscale, eps=1e-8):
    """(demod (B, Cout), wsq (Cout, Cin)) straight from the weight (..., Cout, Cin, kh, kw): the training form of demod_coefs."""
    style, weight = _req(style, "style"), _req(weight, "weight")
    B, Cin = style.shape
    Cout, K = weight.shape[-4], weight.shape[-1] * weight.shape[-2]
    if weight.shape[-3] != Cin or weight.numel() != Cout * Cin * K:
        raise RuntimeError(f"demod_weight: style {tuple(style.shape)} does not match weight {tuple(weight.shape)}")
    out = torch.empty((B, Cout), device=style.device, dtype=torch.float32)
    wsq = torch.empty((Cout, Cin), device=style.device, dtype=torch.float32)
    check(lib.vsp_demod_weight_f32(_ptr(out), _ptr(wsq), _ptr(style), _ptr(weight), B, Cin, Cout, K, float(wscale), float(eps), _stream()),
          "demod_weight")
    return out, wsq


def demod_weight_bwd(g, out, style, wsq, weight, wscale, need_style=True, need_weight=True, ds_out=None, dw_out=None, accumulate=False):
    """Gradient of a loss through demod_weight's `out`: (dstyle (B, Cin) or None, dweight shaped like weight or None).  `g` and `out` may be
    column windows of wider row-major tensors with the SAME row pitch (slices [:, a:b]).  `ds_out` / `dw_out` + `accumulate`: add the
    result to existing contiguous buffers (the convolution's own style / weight gradient) instead of allocating."""
    style, wsq, weight = _req(style, "style"), _req(wsq, "wsq"), _req(weight, "weight")
    B, Cin = style.shape
    Cout, K = wsq.shape[0], weight.numel() // wsq.numel()
    for t, nm in ((g, "g"), (out, "out")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.shape == (B, Cout) and t.stride(1) == 1):
            raise RuntimeError(f"demod_weight_bwd: {nm} must be a float32 device (B, Cout) row-major window")
    if g.stride(0) != out.stride(0) and B > 1:
        raise RuntimeError("demod_weight_bwd: g and out must share their row pitch")
    gs = int(g.stride(0)) if B > 1 else Cout
    ds = (ds_out if ds_out is not None else torch.empty_like(style)) if need_style else None
    dw = (dw_out if dw_out is not None else torch.empty_like(weight)) if need_weight else None
    for t, nm, ref in ((ds, "ds_out", style), (dw, "dw_out", weight)):
        if t is not None and not (t.is_contiguous() and t.numel() == ref.numel() and t.dtype == torch.float32 and t.is_cuda):
            raise RuntimeError(f"demod_weight_bwd: {nm} must be a contiguous float32 device tensor with the operand's element count")
    if accumulate and ((need_style and ds_out is None) or (need_weight and dw_out is None)):
        raise RuntimeError("demod_weight_bwd: accumulate needs the buffers to add to")
    check(lib.vsp_demod_weight_bwd_acc_f32(_ptr(ds), _ptr(dw), C.c_void_p(g.data_ptr()), gs, C.c_void_p(out.data_ptr()), _ptr(style), _ptr(wsq),
                                           _ptr(weight), B, Cin, Cout, K, float(wscale), int(bool(accumulate)), _stream()), "demod_weight_bwd")
    return ds, dw


def avgpool2x2(x):
    x = _req(x, "x")
    H, W = x.shape[-2:]
    if H % 2 or W % 2:
        raise RuntimeError("avgpool2x2 needs even spatial dims")
    out = torch.empty(x.shape[:-2] + (H // 2, W // 2), device=x.device, dtype=x.dtype)
    check(lib.vsp_avgpool2x2_f32(_ptr(out), _ptr(x), x.numel() // (H * W), H // 2, W // 2, _stream()), "avgpool2x2")
    return out


def upsample_add(x, y):
    x, y = _req(x, "x"), _req(y, "y")
    IH, IW = x.shape[-2:]
    OH, OW = y.shape[-2:]
    out = torch.empty_like(y)
    check(lib.vsp_upsample_add_f32(_ptr(out), _ptr(x), _ptr(y), y.numel() // (OH * OW), IH, IW, OH, OW, _stream()),
          "upsample_add")
    return out


def e4e_codes(heads, latent_avg=None):
    """(T, B, D) map2style outputs -> (B, T, D) W+ codes: w0 + delta_i (+ latent_avg), one launch."""
    heads = _req(heads, "heads")
    T, B, D = heads.shape
    out = torch.empty((B, T, D), device=heads.device, dtype=torch.float32)
    check(lib.vsp_e4e_codes_f32(_ptr(out), _ptr(heads), _ptr(_opt(latent_avg, "latent_avg")), B, T, D, _stream()), "e4e_codes")
    return out


def rows_concat(B, T, segments):
    """out (B, T, sum widths) from up to three segments (tensor, width, per_token, flip): per_token=True reads tensor[b, t (or T-1-t),
    :width] of a (B, >= T, >= width) contiguous tensor, per_token=False broadcasts tensor[b, :width] over the tokens."""
    n = len(segments)
    keep = [_req(t, "segment") for t, _, _, _ in segments]
    src = (C.c_void_p * n)(*[t.data_ptr() for t in keep])
    bs = (C.c_int * n)(*[t.stride(0) for t in keep])
    ts = (C.c_int * n)(*[(t.stride(1) if per_tok else 0) for (t, _, per_tok, _) in segments])
    wd = (C.c_int * n)(*[int(w) for _, w, _, _ in segments])
    fl = (C.c_int * n)(*[1 if f else 0 for _, _, _, f in segments])
    out = torch.empty((B, T, sum(int(w) for _, w, _, _ in segments)), device=keep[0].device, dtype=torch.float32)
    check(lib.vsp_rows_concat_f32(_ptr(out), B, T, n, src, bs, ts, wd, fl, _stream()), "rows_concat")
    return out


def resize_bilinear(x, size):
    """F.interpolate(x, size, mode="bilinear", align_corners=False) for NCHW fp32."""
    x = _req(x, "x")
    B, Cc, IH, IW = x.shape
    OH, OW = size
    out = torch.empty((B, Cc, OH, OW), device=x.device, dtype=x.dtype)
    check(lib.vsp_resize_bilinear_f32(_ptr(out), _ptr(x), B * Cc, IH, IW, OH, OW, _stream()), "resize_bilinear")
    return out


def resize_bilinear_bwd(dy, in_size):
    """Adjoint of resize_bilinear: gradient w.r.t. the (IH, IW) input."""
    dy = _req(dy, "dy")
    B, Cc, OH, OW = dy.shape
    IH, IW = in_size
    dx = torch.empty((B, Cc, IH, IW), device=dy.device, dtype=dy.dtype)
    check(lib.vsp_resize_bilinear_bwd_f32(_ptr(dx), _ptr(dy), B * Cc, IH, IW, OH, OW, _stream()), "resize_bilinear_bwd")
    return dx


def maxpool2d(x, k, s, p=0):
    """F.max_pool2d(x, k, s, p) (floor mode) for NCHW fp32."""
    x = _req(x, "x")
    B, Cc, H, W = x.shape
    OH, OW = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    out = torch.empty((B, Cc, OH, OW), device=x.device, dtype=x.dtype)
    check(lib.vsp_maxpool2d_f32(_ptr(out), _ptr(x), B * Cc, H, W, OH, OW, k, s, p, _stream()), "maxpool2d")
    return out


def maxpool2d_bwd(dy, x, k, s, p=0):
    x, dy = _req(x, "x"), _req(dy, "dy")
    B, Cc, H, W = x.shape
    OH, OW = dy.shape[-2:]
    dx = torch.empty_like(x)
    check(lib.vsp_maxpool2d_bwd_f32(_ptr(dx), _ptr(dy), _ptr(x), B * Cc, H, W, OH, OW, k, s, p, _stream()), "maxpool2d_bwd")
    return dx


def lpips_layer(f0, f1, w):
    """One LPIPS level: (B,) = spatial mean of sum_c w_c (unit(f0) - unit(f1))^2 over NCHW fp32 features."""
    f0, f1, w = _req(f0, "f0"), _req(f1, "f1"), _req(w, "w")
    B, Cc, H, W = f0.shape
    if f1.shape != f0.shape or w.numel() != Cc:
        raise RuntimeError("lpips_layer: f0 / f1 / w shapes do not match")
    out = torch.empty((B,), device=f0.device, dtype=torch.float32)
    check(lib.vsp_lpips_layer_f32(_ptr(out), _ptr(f0), _ptr(f1), _ptr(w), B, Cc, H * W, _stream()), "lpips_layer")
    return out


def lpips_layer_bwd(gout, f0, f1, w):
    f0, f1, w, gout = _req(f0, "f0"), _req(f1, "f1"), _req(w, "w"), _req(gout, "gout")
    B, Cc, H, W = f0.shape
    df1 = torch.empty_like(f1)
    check(lib.vsp_lpips_layer_bwd_f32(_ptr(df1), _ptr(f0), _ptr(f1), _ptr(w), _ptr(gout), B, Cc, H * W, _stream()), "lpips_layer_bwd")
    return df1


def plane_mean(x):
    x = _req(x, "x")
    B, Cc, H, W = x.shape
    out = torch.empty((B, Cc), device=x.device, dtype=x.dtype)
    check(lib.vsp_plane_mean_f32(_ptr(out), _ptr(x), B * Cc, H * W, _stream()), "plane_mean")
    return out


def scale_add(x, gate, y=None):
    x = _req(x, "x")
    B, Cc, H, W = x.shape
    out = torch.empty_like(x)
    check(lib.vsp_scale_add_f32(_ptr(out), _ptr(x), _ptr(_req(gate, "gate")), _ptr(_opt(y, "y")), B * Cc, H * W,
                                _stream()), "scale_add")
    return out


def subsample(x, s):
    x = _req(x, "x")
    B, Cc, H, W = x.shape
    out = torch.empty((B, Cc, (H - 1) // s + 1, (W - 1) // s + 1), device=x.device, dtype=x.dtype)
    check(lib.vsp_subsample_f32(_ptr(out), _ptr(x), B * Cc, H, W, s, _stream()), "subsample")
    return out


def add3(a, b, c=None):
    out = torch.empty_like(_req(a, "a"))
    check(lib.vsp_add3_f32(_ptr(out), _ptr(a), _ptr(_req(b, "b")), _ptr(_opt(c, "c")), a.numel(), _stream()), "add3")
    return out


# ----------------------------------------------------------------------------------------------- fused TACC step
def tacc_scores(P, eQ, wq_col, tfrac, B, k_off=0):
    score = torch.empty((B, 18, 18), device=P.device, dtype=P.dtype)
    check(lib.vsp_tacc_scores_f32(_ptr(score), _ptr(_req(P, "P")), P.shape[1], k_off, _ptr(_req(eQ, "eQ")),
                                  C.c_void_p(wq_col.data_ptr()), wq_col.stride(0), float(tfrac), B, 18, 512, _stream()),
          "tacc_scores")
    return score


def tacc_chan_attn(P, ek, wk_col, tfrac, B, q2_off=1024, v2_off=1536):
    t = torch.empty((B, 18, 512), device=P.device, dtype=P.dtype)
    check(lib.vsp_tacc_chan_attn_f32(_ptr(t), _ptr(_req(P, "P")), P.shape[1], q2_off, v2_off, _ptr(_req(ek, "ek")),
                                     C.c_void_p(wk_col.data_ptr()), wk_col.stride(0), float(tfrac), B, 18, 512, _stream()),
          "tacc_chan_attn")
    return t


def tacc_tail(P, eQ, wq, tfrac, t, gamma, beta, B, xold=None, c1=None, c2=None, idx=0, k_off=0, v_off=512, want_pn=True):
    """`wq`: the CONTIGUOUS last weight column of the block's q_matrix (512 floats)."""
    y = torch.empty((B, 18, 512), device=P.device, dtype=P.dtype)
    pn = torch.empty_like(y) if want_pn else None
    check(lib.vsp_tacc_tail_f32(_ptr(y), _ptr(pn), _ptr(_req(P, "P")), P.shape[1], k_off, v_off, _ptr(_req(eQ, "eQ")),
                                _ptr(_req(wq, "wq")), float(tfrac), _ptr(_req(t, "t")), _ptr(_req(gamma, "gamma")),
                                _ptr(_req(beta, "beta")), _ptr(_opt(xold, "xold")), _ptr(_opt(c1, "c1")), _ptr(_opt(c2, "c2")),
                                int(idx), B, 18, 512, _stream()), "tacc_tail")
    return y, pn


def tacc_head_pre(e, wcol, ln_w, ln_b, steps, t_div):
    M = e.shape[0]
    out = torch.empty((steps * M, 512), device=e.device, dtype=e.dtype)
    check(lib.vsp_tacc_head_pre_f32(_ptr(out), _ptr(_req(e, "e")), C.c_void_p(wcol.data_ptr()), wcol.stride(0),
                                    _ptr(_req(ln_w, "ln_w")), _ptr(_req(ln_b, "ln_b")), steps, M, 512, float(t_div), _stream()),
          "tacc_head_pre")
    return out


def pointwise(x, w, in_scale=None, ch_bias=None, bias1=None, bias2=None, res=None, up_src=None, up_kernel=None):
    """1x1 convolution with <= 4 channels on one side as an HBM stream (see vsp_pointwise_f32): x (B,Cin,H,W), w (Cout,Cin);
    bias1 / bias2 switch on the two FusedLeakyReLU stages (few-input form), res is added last (few-output form);
    up_src (B,Cout,H/2,W/2) + up_kernel (4,4): the 2x FIR-upsampled skip is evaluated inside the kernel and added."""
    x = _req(x, "x", bf16_ok=True)
    B, Cin, Hh, Ww = x.shape
    Cout = w.shape[0]
    # bf16 on the WIDE side only: few outputs (ToRGB) read bf16 features and write the fp32 image; few inputs (the 3 -> 64 input
    # layer) read the fp32 image and, in the bf16-activation configuration, write bf16 features
    if Cout > 4 and x.dtype == BF:
        x = to_f32(x)
    wide_bf = (x.dtype == BF) if Cout <= 4 else bool(ACT_BF16 and BF16_CONV is True and min(Hh, Ww) >= 32)
    y = torch.empty((B, Cout, Hh, Ww), device=x.device, dtype=BF if (wide_bf and Cout > 4) else torch.float32)
    fn = lib.vsp_pointwise_bf16 if wide_bf else lib.vsp_pointwise_f32
    check(fn(_ptr(y), _ptr(x), _ptr(_req(w, "w")), _ptr(_opt(in_scale, "in_scale")), _ptr(_opt(ch_bias, "ch_bias")),
             _ptr(_opt(bias1, "bias1")), 1 if bias1 is not None else 0, _ptr(_opt(bias2, "bias2")),
             1 if bias2 is not None else 0, _ptr(_opt(res, "res")), _ptr(_opt(up_src, "up_src")),
             _ptr(_opt(up_kernel, "up_kernel")), Ww, B, Cin, Cout, Hh * Ww, _stream()), "pointwise")
    return y


def tacc_chain(x, blocks, steps, coef_idx=None, c1=None, c2=None, t_div=1.0, head_steps=None):
    """Run the whole sampler chain in place on x (B,18,512): for each t in `steps` (host ints, execution order) x <- c1[k] *
    denoiser(x, t) + c2[k] * x with k = coef_idx[s] (default t); c1 = c2 = None: x <- denoiser(x, t).  `blocks`: one dict per
    TACC block with device tensors wcat, eQ, ek, wq, wk, gamma, beta (gamma/beta: (head_steps, B, 18, 512))."""
    from ._lib import TaccBlock, TaccChainParams
    x = _req(x, "x")
    B = x.shape[0]
    n = len(blocks)
    arr = (TaccBlock * n)()
    keep = []
    for i, blk in enumerate(blocks):
        for name in ("wcat", "eQ", "ek", "wq", "wk", "gamma", "beta"):
            t = _req(blk[name], name)
            keep.append(t)
            setattr(arr[i], name, t.data_ptr())
        if blk.get("wcat_frag") is not None:   # optional: the projection matrix in MFMA fragment order (coalesced loads)
            t = _req(blk["wcat_frag"], "wcat_frag")
            if t.numel() != blk["wcat"].numel():
                raise RuntimeError("tacc_chain: wcat_frag must hold the same elements as wcat")
            keep.append(t)
            arr[i].wcat_frag = t.data_ptr()
    if head_steps is None:
        head_steps = blocks[0]["gamma"].shape[0] if n else 0
    nfl = lib.vsp_tacc_chain_work_floats(B)
    work = torch.empty(nfl, device=x.device, dtype=torch.float32)
    steps = [int(s) for s in steps]
    st = (C.c_int * len(steps))(*steps)
    ci = (C.c_int * len(steps))(*[int(k) for k in coef_idx]) if coef_idx is not None else None
    p = TaccChainParams()
    p.B, p.n_tok, p.dim, p.n_blocks = B, 18, 512, n
    p.blocks = arr
    p.x, p.work, p.work_floats = x.data_ptr(), work.data_ptr(), nfl
    p.n_steps, p.step = len(steps), st
    p.coef_idx = ci
    p.c1 = _opt(c1, "c1").data_ptr() if c1 is not None else None
    p.c2 = _opt(c2, "c2").data_ptr() if c2 is not None else None
    p.t_div, p.head_steps = float(t_div), int(head_steps)
    check(lib.vsp_tacc_chain_f32(C.byref(p), _stream()), "tacc_chain")
    return x


def quantize_u8_nhwc(x, lo=-1.0, hi=1.0):
    """(B, C, H, W) fp32 -> (B, H, W, C) uint8 with torchvision's save_image(normalize=True, value_range=(lo, hi)) rounding."""
    x = _req(x, "x")
    B, Cc, Hh, Ww = x.shape
    out = torch.empty((B, Hh, Ww, Cc), device=x.device, dtype=torch.uint8)
    check(lib.vsp_quantize_u8_nhwc(C.c_void_p(out.data_ptr()), _ptr(x), B, Cc, Hh, Ww, float(lo), float(hi), _stream()),
          "quantize_u8_nhwc")
    return out


# ----------------------------------------------------------------------------------------------- keyed random tensors
# segment ids of the path's draws (they enter the Philox counter: a tensor keeps its values whatever else is drawn with it)
SEG_LQ, SEG_XT, SEG_Z = 1, 2, 3           # synthetic LQ batch (bench), x_T (ldm/ddpm.py:423), z (restoration_test.py:77-82; +1: second mixing code)
SEG_GEN, SEG_ENC, SEG_DEC = 16, 48, 80    # + layer index: prior decoder / Restoration_net encoder / decoder NoiseInjection maps


def keyed_fill(shapes, ids, seed, image_index0, dist="normal", device=None, index_tensor=None):
    """[tensor of shape s for s in shapes], every shape = (B, ...) with the same B, all drawn by ONE launch of
    vsp_keyed_fill_f32 into one allocation: value = f(seed, image_index0 + b, id, element).  dist: "normal" | "uniform"
    (-1, 1).  index_tensor: optional device int64 scalar added to image_index0 at run time (graph replays)."""
    shapes = [tuple(int(d) for d in s) for s in shapes]
    if not shapes:
        return []
    B = shapes[0][0]
    if any(s[0] != B for s in shapes) or len(ids) != len(shapes):
        raise RuntimeError("keyed_fill: every tensor needs the same leading batch dimension and one id")
    elems = [math.prod(s[1:]) for s in shapes]
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if dev.type != "cuda":
        raise RuntimeError("keyed_fill: CUDA(HIP) device required")
    if index_tensor is not None and (index_tensor.dtype != torch.int64 or not index_tensor.is_cuda or index_tensor.numel() != 1):
        raise RuntimeError("keyed_fill: index_tensor must be a device int64 scalar")
    flat = torch.empty(B * sum(elems), device=dev, dtype=torch.float32)
    n = len(shapes)
    check(lib.vsp_keyed_fill_f32(_ptr(flat), B, (C.c_int64 * n)(*elems), (C.c_int32 * n)(*[int(i) for i in ids]), n,
                                 C.c_uint64(int(seed) & (2 ** 64 - 1)), int(image_index0),
                                 None if index_tensor is None else C.c_void_p(index_tensor.data_ptr()), {"normal": 0, "uniform": 1}[dist],
                                 _stream()), "keyed_fill")
    out, off = [], 0
    for s, e in zip(shapes, elems):
        out.append(flat[off:off + B * e].view(s))
        off += B * e
    return out


# ----------------------------------------------------------------------------------------------- convolution backward
def conv2d_wgrad(x, dy, weight_shape, stride=1, padding=0, dilation=1, groups=1, x_scale=None, dy_scale=None, x_shared=False,
                 dy_coff=0, out=None, accumulate=False, scale=1.0):
    """dL/dW of y = conv2d(x * x_scale, W, stride, padding, dilation, groups) * dy_scale given dL/dy (vsp_conv2d_wgrad_f32):
    x (B, G*Cin_g, H, W), dy (B, G*Cout_g, OH, OW) -> (G*Cout_g, Cin_g, KH, KW); the scales are optional (B, channels).
    `x_shared`: every group reads the same Cin_g channels of x; `dilation` / `padding` may then be per-group tuples (<= 4 groups: the
    dilated SMART branches in one launch).  `dy_coff`: first channel of dy used (dy may hold more channels than G*Cout_g).
    `scale` multiplies the result (the layer's equalized-lr factor: the gradient of the parameter, not of the scaled weight)."""
    from ._lib import ConvWgradParams
    x, dy = _req(x, "x"), _req(dy, "dy")
    cout, cin_g, kh, kw = (int(v) for v in weight_shape)
    B, xc, H, W = x.shape
    if dy.shape[0] != B or dy.shape[1] < dy_coff + cout or xc != (cin_g if x_shared else cin_g * groups) or cout % groups:
        raise RuntimeError(f"conv2d_wgrad: x {tuple(x.shape)} / dy {tuple(dy.shape)} do not match weight {tuple(weight_shape)} with {groups} groups")
    dw = out if out is not None else torch.empty((cout, cin_g, kh, kw), device=x.device, dtype=torch.float32)
    if out is not None and (_req(out, "out").shape != (cout, cin_g, kh, kw)):
        raise RuntimeError("conv2d_wgrad: `out` must be (Cout, Cin_g, KH, KW)")
    p = ConvWgradParams()
    keep = [x, dy, dw, _opt(x_scale, "x_scale"), _opt(dy_scale, "dy_scale")]
    p.x, p.dy, p.dw, p.x_scale, p.dy_scale = [(t.data_ptr() if t is not None else None) for t in keep]
    p.B, p.Cin_g, p.H, p.W, p.G, p.Cout_g = B, cin_g, H, W, groups, cout // groups
    p.OH, p.OW, p.KH, p.KW, p.stride = dy.shape[2], dy.shape[3], kh, kw, int(stride)
    per_group = isinstance(dilation, (tuple, list)) or isinstance(padding, (tuple, list))
    if per_group:
        dl = tuple(dilation) if isinstance(dilation, (tuple, list)) else (dilation,) * groups
        pd = tuple(padding) if isinstance(padding, (tuple, list)) else (padding,) * groups
        if len(dl) != groups or len(pd) != groups or groups > 4:
            raise RuntimeError("conv2d_wgrad: per-group dilation / padding need one value per group, at most 4 groups")
        p.per_group_geometry = 1
        for i in range(groups):
            p.dil_g[i], p.pad_g[i] = int(dl[i]), int(pd[i])
        p.dil, p.pad = int(dl[0]), int(pd[0])
        dilation = dl[0]
    else:
        p.dil, p.pad = int(dilation), int(padding)
    p.x_shared, p.dy_ch, p.dy_coff, p.accumulate = int(bool(x_shared)), dy.shape[1], int(dy_coff), int(bool(accumulate))
    p.dw_scale = float(scale)
    # split-K partial sums in private copies of dw (torch's caching allocator: stream-ordered) instead of fp32 atomics
    wf = int(lib.vsp_conv2d_wgrad_work_floats(C.byref(p))) if WGRAD_WORKSPACE else 0
    if wf > 0:
        work = torch.empty((wf,), device=x.device, dtype=torch.float32)
        keep.append(work)
        p.work, p.work_floats = work.data_ptr(), wf
    prof = PROFILER
    start = prof.begin() if prof is not None else None
    check(lib.vsp_conv2d_wgrad_f32(C.byref(p), _stream()), "conv2d_wgrad")
    if prof is not None:
        prof.end(start, 2.0 * B * cout * dy.shape[2] * dy.shape[3] * cin_g * kh * kw,
                 (cin_g, cout // groups, dy.shape[2], dy.shape[3], kh, int(stride), groups, "wgrad", f"d{int(dilation)}"))
    return dw


def plane_dot_scale_(a, b, scale):
    """-> (B, C) = sum over the plane of a * b; `a` (B, C, H, W) is multiplied IN PLACE by scale (B, C)."""
    a, b, scale = _req(a, "a"), _req(b, "b"), _req(scale, "scale")
    if a.shape != b.shape or a.dim() < 3 or scale.numel() != a.shape[0] * a.shape[1]:
        raise RuntimeError("plane_dot_scale_: a, b of the same (B, C, ...) shape and scale (B, C)")
    out = torch.empty(a.shape[:2], device=a.device, dtype=torch.float32)
    planes = a.shape[0] * a.shape[1]
    check(lib.vsp_plane_dot_scale_f32(_ptr(out), _ptr(a), _ptr(b), _ptr(scale), planes, a.numel() // max(planes, 1), _stream()),
          "plane_dot_scale")
    return out


def noise_bias_act(x, noise, noise_w, bias, slope=0.2, gain=SQRT2):
    """lrelu(x + noise_w * noise + bias[c], slope) * gain for x (B, C, H, W), noise (B, 1, H, W), noise_w (1,), bias (C,) or None."""
    x, noise, noise_w = _req(x, "x"), _req(noise, "noise"), _req(noise_w, "noise_w")
    B, Cc, Hh, Ww = x.shape
    if noise.numel() != B * Hh * Ww or noise_w.numel() != 1 or (bias is not None and bias.numel() != Cc):
        raise RuntimeError("noise_bias_act: noise (B, 1, H, W), noise_w (1,), bias (C,)")
    y = torch.empty_like(x)
    check(lib.vsp_noise_bias_act_f32(_ptr(y), _ptr(x), _ptr(noise), _ptr(noise_w), _ptr(_opt(bias, "bias")), B, Cc, Hh * Ww, float(slope),
                                     float(gain), _stream()), "noise_bias_act")
    return y


def noise_dot(gx, noise):
    """(1,) = sum over everything of gx (B, C, H, W) * noise (B, 1, H, W)."""
    gx, noise = _req(gx, "gx"), _req(noise, "noise")
    B, Cc, Hh, Ww = gx.shape
    out = torch.empty((1,), device=gx.device, dtype=torch.float32)
    check(lib.vsp_noise_dot_f32(_ptr(out), _ptr(gx), _ptr(noise), B, Cc, Hh * Ww, _stream()), "noise_dot")
    return out


def smart_tail_bwd(g, y, noise, noise_w, bias2, slope=0.2, gain=SQRT2):
    """Backward of FusedLeakyReLU(bias1) -> NoiseInjection -> FusedLeakyReLU(bias2) from the final output y (vsp_smart_tail_bwd_f32):
    returns (g1 = gradient entering the conv, d bias1 (C,), d bias2 (C,), d noise_weight (1,))."""
    g, y, noise = _req(g, "g"), _req(y, "y"), _req(noise, "noise")
    B, Cc, Hh, Ww = g.shape
    g1 = torch.empty_like(g)
    sums = torch.empty(2 * Cc + 1, device=g.device, dtype=torch.float32)
    check(lib.vsp_smart_tail_bwd_f32(_ptr(g1), _ptr(sums), C.c_void_p(sums.data_ptr() + 4 * Cc), C.c_void_p(sums.data_ptr() + 8 * Cc),
                                     _ptr(g), _ptr(y), _ptr(noise), _ptr(_req(noise_w, "noise_w")), _ptr(_req(bias2, "bias2")), B, Cc,
                                     Hh * Ww, float(slope), float(gain), _stream()), "smart_tail_bwd")
    return g1, sums[:Cc], sums[Cc:2 * Cc], sums[2 * Cc:]


def channel_sum(x):
    """(B, C, ...) -> (C,): sum over the batch and everything behind the channel dimension (bias gradients)."""
    x = _req(x, "x")
    if x.dim() < 2:
        raise RuntimeError("channel_sum: (B, C, ...) tensor")
    B, Cc = x.shape[0], x.shape[1]
    out = torch.empty((Cc,), device=x.device, dtype=torch.float32)
    check(lib.vsp_channel_sum_f32(_ptr(out), _ptr(x), B, Cc, x.numel() // max(B * Cc, 1), _stream()), "channel_sum")
    return out


def plane_dot(a, b):
    """(B, C, H, W) x (B, C, H, W) -> (B, C): sum over the plane of a * b."""
    a, b = _req(a, "a"), _req(b, "b")
    if a.shape != b.shape or a.dim() < 3:
        raise RuntimeError("plane_dot: two tensors of the same (B, C, ...) shape")
    out = torch.empty(a.shape[:2], device=a.device, dtype=torch.float32)
    planes = a.shape[0] * a.shape[1]
    check(lib.vsp_plane_dot_f32(_ptr(out), _ptr(a), _ptr(b), planes, a.numel() // max(planes, 1), _stream()), "plane_dot")
    return out


def _guard_public_ops():
    """every public operator of this module runs under `device_guarded` (helpers without tensor arguments pass straight through)"""
    import types
    skip = {"conv_key", "fir_out_size", "fir_bf16_ok", "conv2d_out_size", "operand_device", "device_guarded", "check"}
    g = globals()
    for name, obj in list(g.items()):
        if isinstance(obj, types.FunctionType) and not name.startswith("_") and name not in skip and obj.__module__ == __name__:
            g[name] = device_guarded(obj)


_guard_public_ops()
