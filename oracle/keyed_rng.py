"""Counter-based, name-keyed random tensors that are bit-identical on every host (test infrastructure).

Weights and noise for parity tests are regenerated from (seed, tensor name) on the build box (reference run,
tools/make_golden.py) and on the GPU box (oracle + HIP run) instead of shipping 1.68 GB of checkpoints.  Only the raw
integer output of numpy's Philox-4x64 bit generator and ONE correctly-rounded fp32 multiply are used, so no libm / SIMD-path
difference can leak in:
    normal-ish(seed, name)[i] = (sum of the four 16-bit fields of h_i - 131070) * c,  c = 1/std   (Irwin-Hall n=4:
    zero mean, unit variance, |x| <= 3.47 -- the tests need O(1) random values, not exact Gaussians)
    uniform(seed, name)[i]    = top 24 bits of h_i * 2^-24   in [0, 1)
"""
import hashlib

import numpy as np
import torch

_IH_STD = float(np.sqrt((65536.0 ** 2 - 1.0) / 3.0))
_IH_SCALE = np.float32(1.0 / _IH_STD)
_CHUNK = 1 << 18  # small chunks stay inside the allocator's heap: fresh mmap pages cost more than the hash


def _bits(seed, name, start, n):
    """n raw 64-bit words [start, start+n) of the Philox-4x64 stream keyed by sha256(seed:name).  Raw BitGenerator
    output is part of numpy's stream-compatibility guarantee; `advance` makes chunks independent of chunk size."""
    d = hashlib.sha256(f"{int(seed)}:{name}".encode()).digest()
    bg = np.random.Philox(key=int.from_bytes(d[:16], "little"))
    if start:
        # one Philox counter step yields 4 words; chunks start on multiples of 4
        assert start % 4 == 0
        bg.advance(start // 4)
    return bg.random_raw(n)


def normal_np(seed, name, shape):
    n = int(np.prod(shape)) if len(shape) else 1
    out = np.empty(n, dtype=np.float32)
    for s in range(0, n, _CHUNK):
        c = min(_CHUNK, n - s)
        h = _bits(seed, name, s, c)
        v = h.view(np.uint16).reshape(c, 4)
        tot = v[:, 0].astype(np.int32) + v[:, 1] + v[:, 2] + v[:, 3]
        out[s:s + c] = (tot - 131070).astype(np.float32) * _IH_SCALE
    return out.reshape(shape)


def uniform_np(seed, name, shape):
    n = int(np.prod(shape)) if len(shape) else 1
    out = np.empty(n, dtype=np.float32)
    for s in range(0, n, _CHUNK):
        c = min(_CHUNK, n - s)
        h = _bits(seed, name, s, c)
        out[s:s + c] = (h >> np.uint64(40)).astype(np.float32) * np.float32(2.0 ** -24)
    return out.reshape(shape)


def normal(seed, name, shape):
    return torch.from_numpy(normal_np(seed, name, tuple(shape)))


def uniform(seed, name, shape, lo=0.0, hi=1.0):
    u = torch.from_numpy(uniform_np(seed, name, tuple(shape)))
    return u * (hi - lo) + lo
