"""One fuzz case (tests/gpu_check.py numbering) through the current library, optionally an older build via SDP_LIB_PATH."""
import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, datagen, parity
want = [int(a) for a in sys.argv[1:]] or [725]
rng = np.random.default_rng(12345)
for it in range(max(want) + 1):
    B = int(rng.integers(1, 5))
    N = int(rng.choice([rng.integers(1, 40), rng.integers(40, 200), rng.integers(200, 700)]))
    M = int(rng.choice([rng.integers(1, 40), rng.integers(40, 200), rng.integers(200, 900)]))
    variant = int(rng.integers(0, 2))
    ts = float(rng.choice([0.1, 1.0, 5.0])); as_ = float(rng.choice([0.1, 1.0, 10.0])); ao = float(rng.choice([0.0, 0.0, 0.5]))
    use_lens = bool(rng.integers(0, 2))
    lens = np.stack([rng.integers(1, N + 1, B), rng.integers(1, M + 1, B)], axis=1).astype(np.int32) if use_lens else None
    if it not in want:
        continue
    theta, A = datagen.theta_A(10000 + it, B, N, M)
    theta = (theta * ts).astype(np.float32); A = (A * as_ + ao).astype(np.float32)
    Z = datagen.normal(20000 + it, (B, N, M))
    if use_lens:
        ref = parity.oracle_lens(theta, A, None, Z, variant, lens); got = parity.engine_all(theta, A, None, Z, variant, lens=lens)
    else:
        Et = (0.5 + datagen.uniform(30000 + it, (B,))).astype(np.float32)
        ref = parity.oracle_all(theta, A, Et, Z, variant); got = parity.engine_all(theta, A, Et, Z, variant)
    print(it, (B, N, M, variant, use_lens, ts, as_, ao), parity.compare(got, ref), "max|Ed ref|", float(np.abs(ref["Ed"]).max()), flush=True)
