"""numpy restatement of vspbfr_amd/csrc/noise.hip (vsp_keyed_fill_f32): Philox4x32-10 keyed by (seed, global image index,
segment id, element) -> Box-Muller normals / uniform(-1,1).  TEST INFRASTRUCTURE: the GPU tests check the kernel against
it (integer stage bit-exact by construction, float stage to libm rounding) and the CPU multi-process test uses it as the
stand-in noise source to show that a sharded batch draws what a single rank draws."""
import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over uint32 arrays c0..c3; k0, k1 scalars.  Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0, k1 = np.uint32(k0 + _W0), np.uint32(k1 + _W1)
    return c0, c1, c2, c3


def _u01(w):
    return ((w >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)


def keyed_fill(shape, seg_id, seed, image_index0, dist="normal"):
    """One tensor (B, ...) as the kernel draws it."""
    B = int(shape[0])
    n = int(np.prod(shape[1:]))
    q = (n + 3) // 4
    e4 = np.tile(np.arange(q, dtype=np.uint32), B)
    img = np.repeat(np.arange(B, dtype=np.uint64) + np.uint64(image_index0), q)
    seed = int(seed) & (2 ** 64 - 1)
    w = philox4x32_10(e4, np.uint32(seg_id), (img & _MASK).astype(np.uint32), (img >> np.uint64(32)).astype(np.uint32),
                      seed & 0xFFFFFFFF, seed >> 32)
    if dist == "normal":
        two_pi = np.float32(6.283185307179586)
        r0 = np.sqrt(np.float32(-2.0) * np.log(_u01(w[0])))
        r1 = np.sqrt(np.float32(-2.0) * np.log(_u01(w[2])))
        t0, t1 = two_pi * _u01(w[1]), two_pi * _u01(w[3])
        v = np.stack([r0 * np.cos(t0), r0 * np.sin(t0), r1 * np.cos(t1), r1 * np.sin(t1)], axis=1)
    else:
        v = np.stack([np.float32(2.0) * _u01(x) - np.float32(1.0) for x in w], axis=1)
    v = v.astype(np.float32).reshape(B, q * 4)[:, :n]
    return np.ascontiguousarray(v).reshape(shape)
