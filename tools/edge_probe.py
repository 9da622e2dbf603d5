"""Parity on shapes outside the regular suite (many strips, many pairs, SW with lengths)."""
import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen, parity
cases = [(1, 5000, 64, 0, False), (1, 64, 2048, 1, False), (1500, 20, 33, 0, False), (700, 70, 65, 1, True), (3, 2100, 130, 0, True), (2, 129, 2047, 1, True)]
for (B, N, M, variant, use_lens) in cases:
    theta, A = datagen.theta_A(9, B, N, M)
    Z = datagen.normal(10, (B, N, M))
    lens = datagen.lengths(11, B, 1, min(N, M)) if use_lens else None
    if use_lens:
        lens[:, 0] = np.minimum(lens[:, 0] * (N // min(N, M)), N); lens[0] = (N, M)
        ref = parity.oracle_lens(theta, A, None, Z, variant, lens)
    else:
        ref = parity.oracle_all(theta, A, None, Z, variant, omp=True)
    got = parity.engine_all(theta, A, None, Z, variant, lens=lens)
    print((B, N, M, variant, use_lens), {k: f"{v:.1e}" for k, v in parity.compare(got, ref).items()}, flush=True)
