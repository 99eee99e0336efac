"""vspbfr_amd -- MI355X (gfx950) implementation of VSPBFR's restoration inference path.

Importing this package loads vspbfr_amd/lib/libvspbfr_hip.so (hand-written HIP kernels behind a C ABI,
include/vspbfr_hip.h) and fails loudly if it is missing: there is no CPU or eager-PyTorch fallback.
"""
from . import _lib  # noqa: F401  (raises ImportError when the HIP library is absent)

__all__ = ["_lib"]
