"""CPU checks of the drop-in boundary: the shared library loads and exports every symbol include/vspbfr_hip.h declares,
and the ctypes structs have the C layout.  No compute calls (no GPU needed)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "vspbfr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vsp_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from vspbfr_amd import _lib
    names = declared_symbols()
    assert len(names) >= 20
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (vsp_[a-z0-9_]+)", out))
    assert set(names) <= exported, sorted(set(names) - exported)
    assert set(names) == set(_lib.exported_symbols()), sorted(set(names) ^ set(_lib.exported_symbols()))


def test_struct_layouts_match():
    import ctypes as C

    from vspbfr_amd import _lib
    for which, st in ((0, _lib.FirEpilogue), (1, _lib.ConvParams), (2, _lib.GemmParams), (3, _lib.TaccBlock),
                      (4, _lib.TaccChainParams)):
        assert _lib.lib.vsp_struct_size(which) == C.sizeof(st)
    assert _lib.lib.vsp_abi_version() == _lib.ABI_VERSION == 4
    assert _lib.lib.vsp_conv2d_num_configs() >= 8
    assert _lib.lib.vsp_conv2d_config_name(0).decode()


def test_invalid_arguments_report_errors_without_a_gpu():
    from vspbfr_amd import _lib
    assert _lib.lib.vsp_conv2d_f32(None, None) == -1
    assert "null params" in _lib.last_error()
    assert _lib.lib.vsp_gemm_f32(None, None) == -1
    # the bf16 entry validates before it touches the device: null params, non-3x3 kernels, stride-2 dilation, Cin % 8
    import ctypes as C
    assert _lib.lib.vsp_conv2d_bf16(None, None) == -1 and "null params" in _lib.last_error()
    p = _lib.ConvParams()
    p.x = p.w = p.y = 256  # non-null, 16-byte aligned dummies: never dereferenced on these paths
    p.B, p.Cin, p.H, p.W, p.G, p.cout_g, p.OH, p.OW = 1, 16, 8, 8, 1, 32, 8, 8
    p.KH = p.KW = 1
    p.stride_y = p.stride_x = p.osy = p.osx = 1
    for g in range(4):
        p.dil[g], p.pad_y[g], p.pad_x[g] = 1, 1, 1
    assert _lib.lib.vsp_conv2d_bf16(C.byref(p), None) == -1 and "3x3" in _lib.last_error()
    p.KH = p.KW = 3
    p.stride_y = p.stride_x = 2
    p.dil[0] = 2
    p.OH = p.OW = 4
    assert _lib.lib.vsp_conv2d_bf16(C.byref(p), None) == -1 and "stride 2" in _lib.last_error()
    p.stride_y = p.stride_x = 1
    p.dil[0], p.OH, p.OW, p.Cin = 1, 8, 8, 12
    assert _lib.lib.vsp_conv2d_bf16(C.byref(p), None) == -1 and "multiple of 8" in _lib.last_error()


def test_product_does_not_import_oracle():
    """The product path may not route through the CPU oracle (tier rule): no module under vspbfr_amd/ mentions it."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "vspbfr_amd")):
        for f in fs:
            if f.endswith(".py") and f != "smoke.py":  # smoke.py is the driver's self-check and uses the oracle as the checker
                txt = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "from oracle" in txt:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_torch_extension_modules_load_and_keep_the_reference_signatures():
    """The AOT pybind11 modules `fused` / `upfirdn2d` (what the reference's `load(...)` calls return, op/fused_act.py:13-20,
    op/upfirdn2d.py:13-20) import without a GPU, expose the reference's function names with its argument counts, and refuse CPU
    tensors the way TORCH_CHECK does (op/fused_bias_act.cpp:10-16)."""
    import pytest
    import torch
    from vspbfr_amd.op import native
    fused, upfirdn2d_op = native.load()
    assert fused.fused_bias_act.__doc__.count("arg") == 7          # input, bias, refer, act, grad, alpha, scale
    assert upfirdn2d_op.upfirdn2d.__doc__.count("arg") == 10       # input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1
    with pytest.raises(RuntimeError):
        fused.fused_bias_act(torch.zeros(2, 3), torch.zeros(3), torch.zeros(0), 3, 0, 0.2, 1.0)
    with pytest.raises(RuntimeError):
        upfirdn2d_op.upfirdn2d(torch.zeros(1, 4, 4, 1), torch.ones(2, 2), 1, 1, 1, 1, 0, 0, 0, 0)


def test_operand_device_refusal_logic():
    """Host side of the device guard (reference op/fused_bias_act.cpp:25, op/upfirdn2d.cpp:23): one device per launch.  CPU tensors stand
    in for devices here; `meta` is a second device type that needs no hardware."""
    import torch
    from vspbfr_amd import hip_ops as H
    a, b = torch.zeros(2), torch.zeros(3)
    assert H.operand_device((a, 1, None, [b, (a,)]), {"k": b}) == torch.device("cpu")
    assert H.operand_device((1, "x"), {}) is None
    m = torch.zeros(2, device="meta")
    with pytest.raises(RuntimeError, match="different devices"):
        H.operand_device((a, m), {})
    with pytest.raises(RuntimeError, match="different devices"):
        H.operand_device((a,), {"res": [m]})
    # every public operator is wrapped, helpers without tensors are not
    assert hasattr(H.fused_bias_act, "__wrapped_op__") and hasattr(H.conv2d_packed, "__wrapped_op__") and hasattr(H.tacc_chain, "__wrapped_op__")
    assert not hasattr(H.conv_key, "__wrapped_op__")
    with pytest.raises(RuntimeError, match="different devices"):
        H.fused_bias_act(a, m, torch.zeros(0), 3, 0, 0.2, 1.0)


def test_tuned_table_winograd_entries_name_eligible_layers():
    """Every shape key of vspbfr_amd/conv_tune.json that names a Winograd kernel describes a layer that kernel's eligibility rule accepts
    (a key the rule refuses would silently fall back at run time and the table's measurement would no longer describe what runs): the
    fused F(4x4) kernel incl. its dilation-group launches (G = 4, dilations 1 / 2 / 4 / 8 over one shared input), the F(4x4) pair, F(2x2)."""
    import json

    from vspbfr_amd import hip_ops as H
    table = json.load(open(os.path.join(ROOT, "vspbfr_amd", "conv_tune.json")))
    seen = {"winograd": 0, "winograd4": 0, "winograd4f": 0}
    for key, val in table.items():
        if val not in seen:
            continue
        parts = key.split(",")
        B, Cin, Hh, Ww, G, cg, kh, kw, stride, d0, OH, OW = (int(v) for v in parts[:12])
        flags = parts[12:]
        assert not any(f.startswith("g") or f in ("q", "t") for f in flags), key     # grouped-input / gradient / transposed layers have no Winograd form
        dil = (1, 2, 4, 8)[:G] if G > 1 else (d0,)
        pc = H.PackedConv(None, G, cg, Cin, kh, kw, stride, dil, dil)
        shift = object() if "s" in flags else None
        if val == "winograd4f":
            ok = H.winograd4f_eligible(pc, Hh, Ww, OH, OW, in_shift=shift)
        elif val == "winograd4":
            ok = H.winograd4_eligible(pc, Hh, Ww, OH, OW, in_shift=shift)
        else:
            ok = H.winograd_eligible(pc, Hh, Ww, OH, OW)
        assert ok, (key, val)
        seen[val] += 1
    assert seen["winograd4f"] >= 20 and seen["winograd4"] >= 3 and seen["winograd"] >= 10, seen
    assert sum(1 for k, v in table.items() if v == "winograd4f" and k.split(",")[4] == "4") >= 8      # the dilation-group launches of DESIGN 6.6


def test_tuning_switches_need_vsp_tune(monkeypatch):
    """Round 6: a stray VSP_* tuning variable must not change anything unless VSP_TUNE=1 is set as well (library: vsp::tune_env)."""
    from vspbfr_amd import _lib
    monkeypatch.delenv("VSP_TUNE", raising=False)
    monkeypatch.setenv("VSP_SOME_SWITCH", "7")
    assert _lib.tune_env("VSP_SOME_SWITCH", "1") == "1"
    monkeypatch.setenv("VSP_TUNE", "0")
    assert _lib.tune_env("VSP_SOME_SWITCH", "1") == "1"
    monkeypatch.setenv("VSP_TUNE", "1")
    assert _lib.tune_env("VSP_SOME_SWITCH", "1") == "7"
    assert _lib.tune_env("VSP_NOT_SET", "x") == "x"
    import os
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vspbfr_amd", "csrc")
    for name in os.listdir(root):   # no kernel source reads the environment directly any more
        if name.endswith((".hip", ".h")) and name != "vsp_common.h":
            src = open(os.path.join(root, name)).read()
            assert not re.search(r"(?<![_:\w])getenv\(", src), name


def test_taps_separable_host_logic():
    """hip_ops.taps_separable: exact outer products of the FLIPPED taps only; the answer rides on the tensor object and follows in-place edits."""
    import torch
    from vspbfr_amd import hip_ops as H
    from vspbfr_amd.layers import make_kernel
    k = make_kernel([1, 3, 3, 1]) * 4
    assert H.taps_separable(k) and k._vsp_separable[1] is True
    g = torch.Generator().manual_seed(1)
    assert not H.taps_separable(torch.randn(4, 4, generator=g))
    u, v = torch.tensor([1.0, 2.0, 4.0, 0.5]), torch.tensor([0.25, 1.0, 3.0, 2.0])
    assert H.taps_separable(torch.outer(u, v))                      # dyadic factors: exact
    k2 = k.clone()
    assert H.taps_separable(k2)
    k2[1, 2] += 0.125                                                # edited in place: re-decided
    assert not H.taps_separable(k2)
    z = k.clone()
    z[3, 3] = 0.0                                                    # the corner the kernels divide by
    assert not H.taps_separable(z)
