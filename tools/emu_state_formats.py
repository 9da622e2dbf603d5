"""CPU emulation: what a smaller saved-state format would cost the first-order result E.
The float64 oracle's weights Q are rounded to a candidate format, the backward recurrence (nw.py:120-135) is run in float64
on the rounded weights, and E is compared with the exact one.  Formats: the shipped 2 x 24-bit fixed point (largest weight
by complement in the reader), 2 x 20-bit, 2 x 16-bit fixed, and "selector + two smaller weights as 15-bit floats" with the
dominant weight by complement (4 bytes per cell).  Data: the benchmark's soft random scores and a peaked case.
usage: python tools/emu_state_formats.py [N M]   (CPU only, ~1 minute)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, datagen
from oracle import oracle

N, M = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 512)


def backward(Q):
    """E from weights Q (N+2, M+2, 3) = (up, diag, left) of each cell, float64 (nw.py:120-135)."""
    n, m = Q.shape[0] - 2, Q.shape[1] - 2
    E = np.zeros((n + 2, m + 2))
    E[n + 1, m + 1] = 1.0
    Q = Q.copy()
    Q[n + 1, m + 1] = 1.0
    for i in range(n, 0, -1):
        for j in range(m, 0, -1):
            E[i, j] = Q[i + 1, j, 0] * E[i + 1, j] + Q[i + 1, j + 1, 1] * E[i + 1, j + 1] + Q[i, j + 1, 2] * E[i, j + 1]
    return E[1:-1, 1:-1]


def fixed(bits):
    def f(Q):
        s = 2.0 ** (bits - 1)   # grid of 2^-(bits-1) like the shipped 24-bit fields (q on a grid of 2^-23)
        R = Q.copy()
        R[..., 0] = np.round(Q[..., 0] * s) / s
        R[..., 2] = np.round(Q[..., 2] * s) / s
        R[..., 1] = np.maximum(1.0 - R[..., 0] - R[..., 2], 0.0)
        return R
    return f


def minifloat(ebits, mbits):
    def rnd(x):
        with np.errstate(divide="ignore"):
            e = np.floor(np.log2(np.where(x > 0, x, 1.0)))
        e = np.maximum(e, -(2 ** ebits - 1))           # smallest exponent: below it the grid stays that of 2^emin (denormals)
        step = 2.0 ** (e - mbits)
        return np.round(x / step) * step
    def f(Q):
        R = Q.copy()
        dom = np.argmax(Q, axis=-1)
        for k in range(3):
            sel = dom != k
            R[..., k] = np.where(sel, rnd(Q[..., k]), 0.0)
        tot = R.sum(axis=-1)
        for k in range(3):
            R[..., k] = np.where(dom == k, np.maximum(1.0 - tot, 0.0), R[..., k])
        return R
    return f


formats = [("2 x 24-bit fixed (shipped, 6 B/cell)", fixed(24)), ("2 x 20-bit fixed (5 B/cell)", fixed(20)), ("2 x 16-bit fixed (4 B/cell)", fixed(16)),
           ("selector + 2 x (5-bit exp, 10-bit mantissa), dominant by complement (4 B/cell)", minifloat(5, 10)),
           ("selector + 2 x (4-bit exp, 11-bit mantissa), dominant by complement (4 B/cell)", minifloat(4, 11)),
           ("selector + 2 x (5-bit exp, 14-bit mantissa) (5 B/cell)", minifloat(5, 14))]
for label, scale in (("soft random scores (bench data)", 1.0), ("peaked (theta x 8)", 8.0)):
    theta, A = datagen.theta_A(77, 1, N, M)
    th, a = (theta * scale).astype(np.float64), A.astype(np.float64)
    Vt, E, Q, Ef = oracle.fwd_bwd(th, a, None, 0, omp=False)
    Q = np.asarray(Q)[0].astype(np.float64)
    E0 = backward(Q)
    print(f"{N} x {M}, {label}: max E {E0.max():.3f}; recurrence check vs oracle {np.abs(E0 - np.asarray(E)[0]).max():.1e}")
    for name, f in formats:
        err = np.abs(backward(f(Q)) - E0).max()
        print(f"   {name:85s} max |dE| = {err:.2e}" + ("   > 1e-4" if err > 1e-4 else ""))
