"""Which input of the adjoint backward sweep carries the Ed error on a long steep Smith-Waterman problem: the oracle's
sweep re-run with the engine's E in place of its own."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, datagen, parity
from oracle import oracle
B, N, M, variant, ts, as_, ao = 2, 2048, 2048, 1, 30.0, 10.0, 0.0
theta, A = datagen.theta_A(81001, B, N, M)
theta = (theta * ts).astype(np.float32); A = (A * as_ + ao).astype(np.float32)
Z = datagen.normal(82001, (B, N, M))
Vt, E, Q, Efull = oracle.fwd_bwd(theta, A, None, variant, omp=True)
Ed, Vtd, Qd = oracle.double_backward(Q, Efull, Z, None, omp=True)
got = parity.engine_all(theta, A, None, Z, variant)
sc = max(1.0, np.abs(Ed).max())
d = np.abs(got["Ed"].astype(np.float64) - Ed)
k = np.unravel_index(np.argmax(d), d.shape)
print("|Ed|max", np.abs(Ed).max(), "worst abs", d.max(), "at", k, "ref", Ed[k], "got", got["Ed"][k], "scaled", d.max() / sc)
print("E there: ref", E[k], "got Ex", got["Ex"][k], " |Ex-E|max", np.abs(got["Ex"] - E).max())
Eh = Efull.copy(); Eh[:, 1:-1, 1:-1] = got["Ex"]
Ed_h = oracle.adjoint_backward(Eh, Q, Qd, omp=True)[:, 1:-1, 1:-1]
print("oracle sweep with the engine's E: scaled err vs oracle", np.abs(Ed_h - Ed).max() / sc, " engine vs that hybrid", np.abs(got["Ed"] - Ed_h).max() / sc)
# error profile along the worst row
b, i, j = k
row = d[b, i]
print("errors along row", i, ": max", row.max(), "cells over 1e-4*sc:", int((d[b] > 1e-4 * sc).sum()), "of", N * M)
# f64 oracle for reference noise
t64, a64, z64 = theta.astype(np.float64), A.astype(np.float64), Z.astype(np.float64)
Vt6, E6, Q6, Ef6 = oracle.fwd_bwd(t64, a64, None, variant, omp=True)
Ed6, Vtd6, _ = oracle.double_backward(Q6, Ef6, z64, None, omp=True)
print("oracle f32 vs f64: Ed scaled", np.abs(Ed6 - Ed).max() / sc, " engine vs f64", np.abs(got["Ed"] - Ed6).max() / sc)
