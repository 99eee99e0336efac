"""The reference's two native modules as AOT pybind11 torch extensions over the C ABI (vspbfr_amd/csrc/torch_ext/, `make torch_ext`):

    fused, upfirdn2d_op = native.load()
    fused.fused_bias_act(input, bias, refer, act, grad, alpha, scale)                                   # op/fused_bias_act.cpp:18-31
    upfirdn2d_op.upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)   # op/upfirdn2d.cpp:17-31

These are what `load("fused", ...)` / `load("upfirdn2d", ...)` return in the reference (op/fused_act.py:13-20, op/upfirdn2d.py:13-20):
replacing those two calls by this import makes the reference's own op/*.py -- autograd Functions included -- run on the gfx950
kernels.  The package itself binds the same library through ctypes (vspbfr_amd/_lib.py) and does not need them."""
import importlib.util
import os
import sysconfig

from .. import _lib  # noqa: F401  (loads libvspbfr_hip.so into the process first: the extensions link against it)

_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lib")


def _load(name):
    path = os.path.join(_DIR, name + sysconfig.get_config_var("EXT_SUFFIX"))
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: build it with `make -C vspbfr_amd/csrc torch_ext`")
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load():
    return _load("fused"), _load("upfirdn2d")
