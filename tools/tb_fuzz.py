"""Fuzz of the batched device traceback against the host walk (deepblast_amd/_dp.py::traceback, the restatement of nw.py:401-444
and of the GPU classes' rule): random shapes around and beyond the 32-cell window, matrices with MANY TIES (small integers),
floor values, bright diagonals, per-pair lengths, both stop rules.  usage: tb_fuzz.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from deepblast_amd._dp import traceback as host_traceback
from deepblast_amd._engine import get_engine
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
eng = get_engine()
n_pairs = n_bad = n_long = 0
for case in range(cases):
    N, M = (int(rng.integers(1, 70)), int(rng.integers(1, 70))) if case % 3 == 0 else (int(rng.integers(30, 420)), int(rng.integers(30, 420)))
    B = 12
    kind = case % 4
    if kind == 0:
        g = rng.integers(0, 3, size=(B, N, M)).astype(np.float32)               # ties everywhere
    elif kind == 1:
        g = rng.normal(size=(B, N, M)).astype(np.float32)
    elif kind == 2:
        g = np.abs(rng.normal(size=(B, N, M))).astype(np.float32)
        g = np.round(g * 2) / 2                                                    # coarse positive values: ties, no early stop
    else:
        g = rng.normal(size=(B, N, M)).astype(np.float32)
        g[rng.random((B, N, M)) < 0.2] = -100000.0 if case % 8 < 4 else -1e10      # the sentinels of both rules
    for b in range(0, B, 3):                                                       # a bright band: long interior walks
        for k in range(min(N, M)):
            g[b, N - 1 - k, M - 1 - k] += 4.0
    lens = np.stack([rng.integers(1, N + 1, B), rng.integers(1, M + 1, B)], axis=1).astype(np.int32)
    for rule in ("cpu", "cuda"):
        for ln in (None, lens):
            st, cn = eng.traceback(torch.from_numpy(g).cuda(), None if ln is None else torch.from_numpy(ln).cuda(), rule=rule)
            st, cn = st.cpu().numpy(), cn.cpu().numpy()
            for b in range(B):
                n, m = (N, M) if ln is None else ln[b]
                n_pairs += 1
                try:
                    want = host_traceback(g[b, :n, :m], rule=rule)
                except IndexError:
                    assert cn[b] == -1, (case, N, M, b, rule)
                    n_bad += 1
                    continue
                assert cn[b] == len(want), (case, N, M, b, rule, cn[b], len(want))
                assert [tuple(int(v) for v in r) for r in st[b, :cn[b]]] == want, (case, N, M, b, rule)
                n_long += len(want) > 64
print(f"{cases} cases, {n_pairs} walks (both rules, with and without lengths): all integer-identical to the host walk; {n_bad} left the matrix (count -1), {n_long} longer than 64 steps")
