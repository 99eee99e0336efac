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
install_loss_networks() (LPIPS / ArcFace goldens only) adds
  4. `torchvision.models.vgg16 / resnet101` from oracle/tv_models.py -- torchvision (requirements.txt:27, 0.13.0) is an un-vendored
     third-party dependency absent here; the two published architectures are restated there -- and empty stand-ins for
     `skimage` / `IPython` (my_lpips/__init__.py:6, networks_basic.py:10-11 import them at module level for metrics the loss
     never calls).
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


def install_loss_networks():
    """torchvision.models (vgg16, resnet101) + skimage / IPython stand-ins for `import my_lpips` and `Loss.id_loss`."""
    install()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import tv_models
    tv = sys.modules.get("torchvision") or types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")
    models.vgg16, models.resnet101 = tv_models.vgg16, tv_models.resnet101
    tv.models = models
    sys.modules["torchvision"], sys.modules["torchvision.models"] = tv, models
    for name, attrs in (("skimage", ()), ("skimage.metrics", ("structural_similarity",)), ("skimage.color", ()),
                        ("skimage.transform", ()), ("IPython", ("embed",))):
        if name not in sys.modules:
            mod = types.ModuleType(name)
            mod.__path__ = []          # importable as a package (`import skimage.transform`)
            for a in attrs:
                setattr(mod, a, None)
            sys.modules[name] = mod
    if not hasattr(sys.modules["IPython"], "get_ipython"):
        sys.modules["IPython"].get_ipython = lambda: None   # matplotlib.pyplot asks once it finds an IPython module
    sys.modules["skimage"].color = sys.modules["skimage.color"]
    sys.modules["skimage"].transform = sys.modules["skimage.transform"]
