"""Deterministic synthetic inputs owned by this repo (no dependence on torch/numpy
RNG stream stability).  A counter-based generator: element k of stream `seed` is
splitmix64(seed * 2^32 + k) mapped to [0,1).  Used by tests, bench.py and
oracle/gen_golden.py so that fixtures generated in the build container and
inputs regenerated on the GPU box are bit-identical.

Distributions follow SURVEY.md 8(d): theta ~ U[0,1), A = -U[0,1), float32.
"""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
    return z ^ (z >> np.uint64(31))


def uniform(seed, shape, dtype=np.float32, offset=0):
    """U[0,1) with 24 random bits per element (exact in float32).  `offset`: first element of the stream to take --
    uniform(s, (hi - lo,) + tail, offset=lo * prod(tail)) is rows lo..hi-1 of uniform(s, (B,) + tail)."""
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        k = np.arange(n, dtype=np.uint64) + np.uint64(offset) + (np.uint64(seed) << np.uint64(32))
        bits = _splitmix64(k) >> np.uint64(40)
    return (bits.astype(np.float64) * (1.0 / (1 << 24))).astype(dtype).reshape(shape)


def normal(seed, shape, dtype=np.float32):
    """Approximately N(0,1): sum of 4 uniforms, centred and scaled (cheap, deterministic)."""
    u = sum(uniform(seed * 4 + i + 1000003, shape, np.float64) for i in range(4))
    return ((u - 2.0) * np.sqrt(3.0)).astype(dtype)


def theta_A(seed, B, N, M, dtype=np.float32, rows=None):
    """rows=(lo, hi): only pairs lo..hi-1 of the (B, N, M) batch (a rank's shard, without generating the rest)."""
    lo, hi = (0, B) if rows is None else rows
    theta = uniform(2 * seed, (hi - lo, N, M), dtype, offset=lo * N * M)
    A = (-uniform(2 * seed + 1, (hi - lo, N, M), dtype, offset=lo * N * M)).astype(dtype)
    return theta, A


def lengths(seed, B, lo, hi):
    """(B,2) int32 lengths uniform in [lo,hi]."""
    u = uniform(seed + 77, (B, 2), np.float64)
    return (lo + np.floor(u * (hi - lo + 1))).astype(np.int32)
