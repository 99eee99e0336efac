"""upfirdn2d: API mirror of reference op/upfirdn2d.py:346-362 over the gfx950 kernel (forward only)."""
from collections import abc

from .. import hip_ops


class _UpfirdnModule:
    """Stand-in for the pybind11 module `upfirdn2d` of the reference (op/upfirdn2d.py:13-20, op/upfirdn2d.cpp:17-31)."""

    @staticmethod
    def upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1):
        return hip_ops.upfirdn2d_native_layout(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1, pad_y0, pad_y1)


upfirdn2d_op = _UpfirdnModule()


def upfirdn2d(input, kernel, up=1, down=1, pad=(0, 0)):
    if not isinstance(up, abc.Iterable):
        up = (up, up)
    if not isinstance(down, abc.Iterable):
        down = (down, down)
    if len(pad) == 2:
        pad = (pad[0], pad[1], pad[0], pad[1])
    if input.device.type != "cuda":
        raise RuntimeError("vspbfr_amd.op.upfirdn2d: input must be a CUDA (HIP) tensor; the CPU statement of this op "
                           "lives in oracle/ and is test-only")
    batch, channel, in_h, in_w = input.shape
    out = upfirdn2d_op.upfirdn2d(input.reshape(-1, in_h, in_w, 1).contiguous(), kernel.contiguous(), up[0], up[1],
                                 down[0], down[1], pad[0], pad[1], pad[2], pad[3])
    return out.view(-1, channel, out.shape[1], out.shape[2])
