"""Mirror of the reference's `op` package for the inference path (reference op/__init__.py:2-5):
`from vspbfr_amd.op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d, conv2d_gradfix`.
The ops run hand-written gfx950 kernels through the C ABI and are differentiable to any order when autograd is enabled
(every derivative is again one of the same kernels: fused_act.py, upfirdn2d.py, conv2d_gradfix.py)."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu, fused  # noqa: F401
from .upfirdn2d import upfirdn2d, upfirdn2d_op  # noqa: F401
from . import conv2d_gradfix  # noqa: F401
