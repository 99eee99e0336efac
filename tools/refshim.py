"""In-process shims that let the read-only reference tree (/root/reference) be imported on a CPU-only host.

BUILD-CONTAINER ONLY.  Nothing under tests/ -m gpu, bench.py or smoke() may import this file: the reference
does not exist on the GPU box.  It is used by tools/make_golden.py (fixture generation) and tools/time_reference_cpu.py.

The three shims, none of which edits a reference file:
  1. torch.utils.cpp_extension.load -> stub (the JIT build of op/*.cu needs nvcc; CPU tensors never reach the
     extension: op/fused_act.py:217, op/upfirdn2d.py:356), and os.makedirs made a no-op under the read-only tree
     (op/fused_act.py:11-12).
  2. a dummy `cv2` module (op/__init__.py:6 -> op/utils.py:6).
  3. sys.modules aliases op.fused_act_cpu / op.upfirdn2d_cpu -> op.fused_act / op.upfirdn2d
     (e4e/models/stylegan2/model.py:7-12 imports them when torch.cuda.is_available() is False).
"""
import os
import sys
import types

REF_ROOT = os.environ.get("VSPBFR_REFERENCE", "/root/reference")


def install():
    if getattr(install, "_done", False):
        return
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"reference tree not found at {REF_ROOT}; this tool only runs in the build container")
    import torch.utils.cpp_extension as cpp_ext

    class _Stub:
        def __getattr__(self, name):
            raise RuntimeError("reference CUDA extension is stubbed on this host (CPU tensors only)")

    cpp_ext.load = lambda *a, **k: _Stub()

    real_makedirs = os.makedirs

    def makedirs(path, *a, **k):
        if os.path.abspath(str(path)).startswith(os.path.abspath(REF_ROOT)):
            return None
        return real_makedirs(path, *a, **k)

    os.makedirs = makedirs

    for name in ("cv2", "matplotlib", "matplotlib.pyplot"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                mod = types.ModuleType(name)
                mod.use = lambda *a, **k: None
                sys.modules[name] = mod
    sys.path.insert(0, REF_ROOT)
    import op  # noqa: F401  (op/__init__.py rebinds op.upfirdn2d to the function, so go through sys.modules)
    sys.modules["op.fused_act_cpu"] = sys.modules["op.fused_act"]
    sys.modules["op.upfirdn2d_cpu"] = sys.modules["op.upfirdn2d"]
    install._done = True
