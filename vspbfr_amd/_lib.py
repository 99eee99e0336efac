"""ctypes binding of libvspbfr_hip.so (the C ABI declared in include/vspbfr_hip.h).

The product path has NO fallback: if the shared library is missing or does not export the ABI this module raises at
import time, and every wrapper raises RuntimeError when the library reports an error (the reference raises
RuntimeError from TORCH_CHECK, op/fused_bias_act.cpp:10-16).
"""
import ctypes as C
import os

# Load order matters: the host program's HIP runtime (the libamdhip64.so.7 that PyTorch-ROCm bundles) must be in the
# process BEFORE this library is dlopen'ed, so that libvspbfr_hip.so binds to it by SONAME and shares its streams and
# allocations.  Loading this library first would pull /opt/rocm's runtime in and leave two HIP runtimes in one process.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VSPBFR_HIP_LIB", os.path.join(_HERE, "lib", "libvspbfr_hip.so"))



def tune_env(name, default=None):
    """A/B and tuning switches (VSP_*) are honoured only under VSP_TUNE=1 (the library's vsp::tune_env has the same gate): a stray variable
    in a production environment cannot change which kernel runs.  VSPBFR_HIP_LIB / VSPBFR_CONV_TUNE / VSPBFR_CONV_DTYPE are configuration,
    not tuning, and are always read."""
    if os.environ.get("VSP_TUNE", "0") in ("", "0"):
        return default
    return os.environ.get(name, default)


ABI_VERSION = 4  # include/vspbfr_hip.h VSP_ABI_VERSION
c_float_p = C.c_void_p  # device pointers are passed as integers (tensor.data_ptr())


class FirEpilogue(C.Structure):
    _fields_ = [
        ("plane_scale", C.c_void_p), ("noise", C.c_void_p), ("noise_w", C.c_void_p), ("act_bias", C.c_void_p),
        ("res1", C.c_void_p), ("res2", C.c_void_p),
        ("channels", C.c_int), ("act", C.c_int), ("slope", C.c_float), ("gain", C.c_float), ("flags", C.c_int),
    ]


FIR_SEPARABLE = 1   # vsp_fir_epilogue.flags: the taps are an outer product (include/vspbfr_hip.h VSP_FIR_SEPARABLE)


class ConvParams(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("y", C.c_void_p),
        ("B", C.c_int), ("Cin", C.c_int), ("H", C.c_int), ("W", C.c_int),
        ("G", C.c_int), ("cout_g", C.c_int),
        ("OH", C.c_int), ("OW", C.c_int),
        ("KH", C.c_int), ("KW", C.c_int),
        ("stride_y", C.c_int), ("stride_x", C.c_int),
        ("dil", C.c_int * 4), ("pad_y", C.c_int * 4), ("pad_x", C.c_int * 4),
        ("y_ch", C.c_int), ("y_coff", C.c_int), ("y_h", C.c_int), ("y_w", C.c_int),
        ("osy", C.c_int), ("osx", C.c_int), ("ooy", C.c_int), ("oox", C.c_int),
        ("in_scale", C.c_void_p), ("in_scale_bstride", C.c_int), ("in_shift", C.c_void_p),
        ("out_scale", C.c_void_p), ("ch_scale", C.c_void_p), ("ch_bias", C.c_void_p),
        ("act1", C.c_int), ("bias1", C.c_void_p), ("slope1", C.c_float), ("gain1", C.c_float),
        ("noise", C.c_void_p), ("noise_w", C.c_void_p),
        ("act2", C.c_int), ("bias2", C.c_void_p), ("prelu", C.c_void_p), ("slope2", C.c_float), ("gain2", C.c_float),
        ("res1", C.c_void_p), ("res2", C.c_void_p), ("res_ch", C.c_int), ("res_coff", C.c_int),
        ("tile_hint", C.c_int), ("x_ch", C.c_int), ("x_group_stride", C.c_int), ("transposed", C.c_int),
        ("io_bf16", C.c_int),
        ("dil_by_input_quarter", C.c_int),
        ("w_bstride", C.c_int64),
    ]


class ConvWgradParams(C.Structure):
    _fields_ = [("x", C.c_void_p), ("dy", C.c_void_p), ("dw", C.c_void_p), ("x_scale", C.c_void_p), ("dy_scale", C.c_void_p)] + [
        (n, C.c_int) for n in ("B", "Cin_g", "H", "W", "G", "Cout_g", "OH", "OW", "KH", "KW", "stride", "dil", "pad", "x_ch", "x_coff", "dy_ch",
                               "dy_coff", "x_shared", "per_group_geometry")] + [("dil_g", C.c_int * 4), ("pad_g", C.c_int * 4),
                                                                                ("accumulate", C.c_int), ("work", C.c_void_p),
                                                                                ("work_floats", C.c_size_t), ("dw_scale", C.c_float)]


class TaccBlock(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("wcat", "eQ", "ek", "wq", "wk", "gamma", "beta", "wcat_frag")]


class TaccChainParams(C.Structure):
    _fields_ = [
        ("B", C.c_int), ("n_tok", C.c_int), ("dim", C.c_int), ("n_blocks", C.c_int),
        ("blocks", C.POINTER(TaccBlock)), ("x", C.c_void_p), ("work", C.c_void_p), ("work_floats", C.c_size_t),
        ("n_steps", C.c_int), ("step", C.POINTER(C.c_int)), ("coef_idx", C.POINTER(C.c_int)),
        ("c1", C.c_void_p), ("c2", C.c_void_p), ("t_div", C.c_float), ("head_steps", C.c_int),
    ]


class GemmParams(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("Bm", C.c_void_p), ("C", C.c_void_p),
        ("Z", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("a_zs", C.c_int64), ("a_ms", C.c_int64), ("a_ks", C.c_int64),
        ("b_zs", C.c_int64), ("b_ns", C.c_int64), ("b_ks", C.c_int64),
        ("c_zs", C.c_int64), ("c_ms", C.c_int64),
        ("alpha", C.c_float), ("bias", C.c_void_p), ("bias_scale", C.c_float),
        ("act", C.c_int), ("slope", C.c_float), ("gain", C.c_float), ("bias_zs", C.c_int64),
    ]


_i, _i64, _f, _p = C.c_int, C.c_int64, C.c_float, C.c_void_p

# name -> argtypes: every symbol include/vspbfr_hip.h declares (tests/test_abi.py cross-checks against the header)
SIGNATURES = {
    "vsp_abi_version": [],
    "vsp_device_count": [],
    "vsp_struct_size": [_i],
    "vsp_fused_bias_act_f32": [_p, _p, _p, _p, _i64, _i, _i, _i, _i, _f, _f, _p],
    "vsp_upfirdn2d_f32": [_p, _p, _p] + [_i] * 14 + [C.POINTER(FirEpilogue), _p],
    "vsp_conv2d_f32": [C.POINTER(ConvParams), _p],
    "vsp_conv2d_num_configs": [],
    "vsp_gemm_f32": [C.POINTER(GemmParams), _p],
    "vsp_pixelnorm_dim1_f32": [_p, _p, _i, _i, _i, _f, _p],
    "vsp_layernorm_f32": [_p, _p, _p, _p, _p, _i, _i, _f, _i, _f, _f, _p],
    "vsp_softmax_lastdim_f32": [_p, _p, _i, _i, _p],
    "vsp_softmax_dim1_f32": [_p, _p, _i, _i, _i, _p],
    "vsp_film_f32": [_p, _p, _p, _p, _i64, _p],
    "vsp_axpby_idx_f32": [_p, _p, _p, _p, _p, _i, _i64, _p],
    "vsp_demod_f32": [_p, _p, _p, _i, _i, _i, _f, _f, _p],
    "vsp_style_plan_f32": [_p, _i, _p, _i, _i64, _i, _i, _i, _f, _p],
    "vsp_demod_weight_f32": [_p, _p, _p, _p, _i, _i, _i, _i, _f, _f, _p],
    "vsp_demod_weight_bwd_f32": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p],
    "vsp_demod_weight_bwd_acc_f32": [_p, _p, _p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p],
    "vsp_avgpool2x2_f32": [_p, _p, _i64, _i, _i, _p],
    "vsp_upsample_add_f32": [_p, _p, _p, _i64, _i, _i, _i, _i, _p],
    "vsp_e4e_codes_f32": [_p, _p, _p, _i, _i, _i, _p],
    "vsp_rows_concat_f32": [_p, _i, _i, _i, C.POINTER(C.c_void_p), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i), _p],
    "vsp_resize_bilinear_f32": [_p, _p, _i64, _i, _i, _i, _i, _p],
    "vsp_plane_mean_f32": [_p, _p, _i64, _i, _p],
    "vsp_scale_add_f32": [_p, _p, _p, _p, _i64, _i, _p],
    "vsp_subsample_f32": [_p, _p, _i64, _i, _i, _i, _p],
    "vsp_add3_f32": [_p, _p, _p, _p, _i64, _p],
    "vsp_pointwise_f32": [_p, _p, _p, _p, _p, _p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i64, _p],
    "vsp_quantize_u8_nhwc": [_p, _p, _i, _i, _i, _i, _f, _f, _p],
    "vsp_tacc_scores_f32": [_p, _p, _i, _i, _p, _p, _i, _f, _i, _i, _i, _p],
    "vsp_tacc_chan_attn_f32": [_p, _p, _i, _i, _i, _p, _p, _i, _f, _i, _i, _i, _p],
    "vsp_tacc_tail_f32": [_p, _p, _p, _i, _i, _i, _p, _p, _f, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "vsp_tacc_head_pre_f32": [_p, _p, _p, _i, _p, _p, _i, _i, _i, _f, _p],
    "vsp_tacc_chain_f32": [_p, _p],
    "vsp_conv2d_winograd_f32": [_p, _p],
    "vsp_conv2d_bf16": [_p, _p],
    "vsp_conv2d_bf16x3": [_p, _p],
    "vsp_conv2d_bf16rv": [_p, _p],
    "vsp_conv2d_bf16dg": [_p, _p],
    "vsp_affine_sample_f32": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "vsp_affine_sample_bwd_f32": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "vsp_color_affine_f32": [_p, _p, _p, _p, _i, _i64, _p],
    "vsp_maxpool2d_f32": [_p, _p, _i64, _i, _i, _i, _i, _i, _i, _i, _p],
    "vsp_maxpool2d_bwd_f32": [_p, _p, _p, _i64, _i, _i, _i, _i, _i, _i, _i, _p],
    "vsp_lpips_layer_f32": [_p, _p, _p, _p, _i, _i, _i, _p],
    "vsp_lpips_layer_bwd_f32": [_p, _p, _p, _p, _p, _i, _i, _i, _p],
    "vsp_resize_bilinear_bwd_f32": [_p, _p, _i64, _i, _i, _i, _i, _p],
    "vsp_conv2d_wgrad_f32": [C.POINTER(ConvWgradParams), _p],
    "vsp_plane_dot_f32": [_p, _p, _p, _i64, _i64, _p],
    "vsp_channel_sum_f32": [_p, _p, _i, _i, _i64, _p],
    "vsp_noise_bias_act_f32": [_p, _p, _p, _p, _p, _i, _i, _i64, C.c_float, C.c_float, _p],
    "vsp_noise_dot_f32": [_p, _p, _p, _i, _i, _i64, _p],
    "vsp_smart_tail_bwd_f32": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i64, C.c_float, C.c_float, _p],
    "vsp_plane_dot_scale_f32": [_p, _p, _p, _p, _i64, _i64, _p],
    "vsp_convert_f32_to_bf16": [_p, _p, _i64, _p],
    "vsp_convert_bf16_to_f32": [_p, _p, _i64, _p],
    "vsp_upfirdn2d_bf16": [_p, _p, _p] + [_i] * 14 + [C.POINTER(FirEpilogue), _p],
    "vsp_pointwise_bf16": [_p, _p, _p, _p, _p, _p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i64, _p],
    "vsp_keyed_fill_f32": [_p, _i, C.POINTER(C.c_int64), C.POINTER(C.c_int32), _i, C.c_uint64, _i64, _p, _i, _p],
    "vsp_conv1x1_small_f32": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _f, _p],
    "vsp_pack_weight_f32": [_p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _p],
    "vsp_winograd_weight_f32": [_p, _p, _i, _i, _i, _p],
    "vsp_winograd4_weight_f32": [_p, _p, _i, _i, _p],
    "vsp_conv2d_winograd4_f32": [_p, _p, C.c_size_t, _p],
    "vsp_winograd4f_weight_f32": [_p, _p, _i, _i, _p],
    "vsp_conv2d_winograd4f_f32": [_p, _p],
    "vsp_modulate_weight_bf16": [_p, _p, _p, _i, _i64, _i, _i, _i, _p],
    "vsp_conv2d_winograd_chunk": [],
    "vsp_conv2d_winograd_mbw": [_i],
}
_CHARP = {"vsp_last_error": [], "vsp_conv2d_config_name": [_i]}
_SIZET = {"vsp_tacc_chain_work_floats": [_i], "vsp_conv2d_wgrad_work_floats": [C.POINTER(ConvWgradParams)],
          "vsp_winograd_weight_floats": [_i, _i, _i], "vsp_winograd4_weight_floats": [_i, _i], "vsp_winograd4f_weight_floats": [_i, _i],
          "vsp_conv2d_winograd4_work_floats": [_p], "vsp_modulate_weight_bf16_bytes": [_i, _i, _i]}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"vspbfr_amd: HIP library not found at {LIB_PATH}. Build it with `python -c 'import __graft_entry__ as g; "
            f"g.build()'` or `make -C vspbfr_amd/csrc`; there is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = ABI mismatch: fail loudly
        fn.argtypes = args
        fn.restype = C.c_int
    for name, args in _CHARP.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_char_p
    for name, args in _SIZET.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_size_t
    if lib.vsp_abi_version() != ABI_VERSION:
        raise ImportError(f"vspbfr_amd: ABI version {lib.vsp_abi_version()} != {ABI_VERSION}")
    for which, st in ((0, FirEpilogue), (1, ConvParams), (2, GemmParams), (3, TaccBlock), (4, TaccChainParams),
                      (5, ConvWgradParams)):
        if lib.vsp_struct_size(which) != C.sizeof(st):
            raise ImportError(f"vspbfr_amd: struct layout mismatch for {st.__name__}: "
                              f"C {lib.vsp_struct_size(which)} vs ctypes {C.sizeof(st)}")
    return lib


lib = _load()


def last_error():
    return lib.vsp_last_error().decode("utf-8", "replace")


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")


def exported_symbols():
    return sorted(list(SIGNATURES) + list(_CHARP) + list(_SIZET))
