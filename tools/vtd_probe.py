"""Where the directional derivative Vtd loses digits on steep scores: per-pair error of the adjoint forward sweep
for one full batch, default build against forced wave counts, and against an fp64 re-evaluation of the same state."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen, parity
from deepblast_amd._engine import get_engine
B, N, M = 129, 348, 303
ts, as_ = 30.0, 10.0
theta, A = datagen.theta_A(50046, B, N, M)
theta = (theta * ts).astype(np.float32); A = (A * as_).astype(np.float32)
Z = datagen.normal(60046, (B, N, M))
ref = parity.oracle_all(theta, A, None, Z, 0)
eng = get_engine()
for waves in (0, 8, 4, 2):
    eng.force_waves = {} if not waves else {p: waves for p in ("fwd", "bwd", "afwd", "abwd", "fwd_x", "bwd_x")}
    try:
        got = parity.engine_all(theta, A, None, Z, 0)
    except Exception as ex:
        print("waves", waves, "failed:", ex); continue
    d = np.abs(got["Vtd"].astype(np.float64) - ref["Vtd"]) / np.maximum(1, np.abs(ref["Vtd"]))
    k = int(np.argmax(d))
    print(f"waves={waves}: Vtd worst rel {d.max():.3e} at pair {k} (ref {ref['Vtd'][k]:.6f} got {got['Vtd'][k]:.6f}), "
          f"pairs over 5e-5: {(d > 5e-5).sum()}, Vt err {parity.rel_err(got['Vt'], ref['Vt']):.2e}, Ed {parity.abs_err(got['Ed'], ref['Ed'], scale=True):.2e}")
