"""First- and second-order errors on large steep problems (long saturated paths), one line per case; the oracle in
float64 beside it: where the reference's own fp32 rounding is the larger term, `ref32-64` shows it."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, datagen, parity
cases = [(1, 8192, 2048, 0, 30.0, 1.0, 0.0), (1, 20000, 2048, 0, 30.0, 1.0, 0.0), (1, 20000, 2048, 1, 8.0, 1.0, 0.0),
         (1, 20000, 2048, 0, 1.0, 1.0, 0.0), (1, 60000, 1000, 0, 30.0, 10.0, 0.0), (2, 2048, 2048, 0, 30.0, 1.0, 0.0)]
if len(sys.argv) > 1: cases = [tuple(float(v) if "." in v else int(v) for v in a.split(",")) for a in sys.argv[1:]]
for it, (B, N, M, variant, ts, as_, ao) in enumerate(cases):
    theta, A = datagen.theta_A(83000 + it, B, N, M)
    theta = (theta * ts).astype(np.float32); A = (A * as_ + ao).astype(np.float32)
    Z = datagen.normal(84000 + it, (B, N, M))
    ref = parity.oracle_all(theta, A, None, Z, variant)
    r64 = parity.oracle_all(theta.astype(np.float64), A.astype(np.float64), None, Z.astype(np.float64), variant)
    got = parity.engine_all(theta, A, None, Z, variant)
    e, e64, n = parity.compare(got, ref), parity.compare(got, r64), parity.compare(ref, r64)
    print((B, N, M, variant, ts, as_, ao), flush=True)
    for nm, d in (("  vs ref fp32", e), ("  vs ref fp64", e64), ("  ref32-64   ", n)):
        print(nm, " ".join(f"{k}={v:.2e}" for k, v in d.items()), flush=True)
