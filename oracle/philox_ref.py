"""ORACLE (test infrastructure only -- never imported by the product path).

numpy restatement of the device-side Gaussian stream that ``spdm_sample`` uses
when the caller passes no pre-drawn noise (include/spdm.h).  This stream is an
EXTENSION: the reference draws its per-step noise with the global torch RNG
inside ``DDPMScheduler.step`` (called at
``/root/reference/models/diffusion_ddpm.py:274``), which no device kernel can
reproduce, so fixed-noise parity runs pass the noise tensor explicitly and
this generator is only checked against itself (integer part bit-exact,
Box-Muller output to fp32 tolerance).

Philox4x32-10 (Salmon et al., SC'11).  counter = (q, sample, step, 0) where
``q`` = flat element index // 4 inside one (H, D) trajectory, ``sample`` is the
GLOBAL trajectory index (so the stream does not depend on how the batch is
sharded over ranks) and ``step`` the loop iteration; key = (seed lo, seed hi).
The four output words give four normals: (w0, w1) -> z0, z1 and (w2, w3) ->
z2, z3 via Box-Muller with u = ((w >> 8) + 0.5) * 2**-24.
"""
from __future__ import annotations

import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint32) for v in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    mask = np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for r in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & mask).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & mask).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def _u01(w):
    return ((w >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(2.0 ** -24)


def normal_words(w0, w1, w2, w3):
    out = []
    for a, b in ((w0, w1), (w2, w3)):
        u1, u2 = _u01(a), _u01(b)
        r = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
        ang = (np.float32(6.283185307179586) * u2).astype(np.float32)
        out += [(r * np.cos(ang)).astype(np.float32), (r * np.sin(ang)).astype(np.float32)]
    return out


def step_noise(seed: int, step: int, sample_offset: int, batch: int, elems: int) -> np.ndarray:
    """Noise (batch, elems) for loop iteration ``step``; trajectory b of this
    shard is global trajectory ``sample_offset + b``."""
    nq = (elems + 3) // 4
    q = np.arange(nq, dtype=np.uint32)[None, :]
    s = (np.arange(batch, dtype=np.uint64) + np.uint64(sample_offset)).astype(np.uint32)[:, None]
    w = philox4x32_10(q, s, np.uint32(step), np.uint32(0),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    z = np.stack(normal_words(*w), axis=-1).reshape(batch, nq * 4)
    return z[:, :elems]
