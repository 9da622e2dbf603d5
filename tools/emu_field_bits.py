"""Gate of the packed field width (round 5): the float64 oracle's weights rounded to a grid of 2^-bits (bits = 20 .. 16),
the backward recurrence (nw.py:120-135) run in float64 on them, max |dE| against the exact one over three seeds and five score
families.  usage: python tools/emu_field_bits.py N M   (CPU only; output kept in profiles/r05_emu_state_formats.txt)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np, datagen
from oracle import oracle
N, M = int(sys.argv[1]), int(sys.argv[2])
def backward(Q):
    n, m = Q.shape[0] - 2, Q.shape[1] - 2
    E = np.zeros((n + 2, m + 2)); E[n + 1, m + 1] = 1.0
    Q = Q.copy(); Q[n + 1, m + 1] = 1.0
    # vectorised over anti-diagonals
    for d in range(n + m, 1, -1):
        i = np.arange(max(1, d - m), min(n, d - 1) + 1); j = d - i
        E[i, j] = Q[i + 1, j, 0] * E[i + 1, j] + Q[i + 1, j + 1, 1] * E[i + 1, j + 1] + Q[i, j + 1, 2] * E[i, j + 1]
    return E[1:-1, 1:-1]
def grid(bits):
    def f(Q):
        s = 2.0 ** bits
        R = Q.copy()
        R[..., 0] = np.round(Q[..., 0] * s) / s
        R[..., 2] = np.round(Q[..., 2] * s) / s
        R[..., 1] = np.maximum(1.0 - R[..., 0] - R[..., 2], 0.0)
        return R
    return f
cases = [("bench", 1.0, 0.0, 1.0, 0.0), ("theta x8", 8.0, 0.0, 1.0, 0.0), ("theta x12, A x6", 12.0, 0.0, 6.0, 0.0), ("theta x4", 4.0, 0, 1, 0), ("theta x30", 30.0, 0, 1, 0)]
for label, ts, to, as_, ao in cases:
    worst = {b: 0.0 for b in (20, 19, 18, 17, 16)}
    for seed in (77, 78, 79):
        theta, A = datagen.theta_A(seed, 1, N, M)
        th, a = (theta * ts + to).astype(np.float64), (A * as_ + ao).astype(np.float64)
        Vt, E, Q, Ef = oracle.fwd_bwd(th, a, None, 0, omp=False)
        Q = np.asarray(Q)[0].astype(np.float64)
        E0 = backward(Q)
        for b in worst:
            worst[b] = max(worst[b], np.abs(backward(grid(b)(Q)) - E0).max())
    print(f"{N}x{M} {label:18s} " + "  ".join(f"{b}-bit {v:.2e}" for b, v in worst.items()), flush=True)
