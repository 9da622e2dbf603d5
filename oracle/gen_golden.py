#!/usr/bin/env python
"""Generate tests/golden/*.npz by running the REAL reference
(/root/reference/deepblast/{nw,sw}.py) in this container.

The reference needs numba; oracle/_shim/numba is a no-op decorator stand-in so the
@njit bodies run as plain numpy/python (same arithmetic, float64 internals).
Only this script touches /root/reference; the fixtures it writes are data (inputs
and the reference's outputs), committed, and are what travels to the GPU box.

Run:  python oracle/gen_golden.py            (about 2 minutes)
"""
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "_shim"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tests"))

from deepblast.nw import NeedlemanWunschDecoder  # noqa: E402
from deepblast.sw import SmithWatermanDecoder  # noqa: E402
import datagen  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
DEC = {"nw": NeedlemanWunschDecoder, "sw": SmithWatermanDecoder}


def run_ref(kind, theta, A, Et=None, Z=None, want_second=True):
    """Reference forward + backward (+ double backward for cotangent Z on theta.grad)."""
    dec = DEC[kind]("softmax")
    t = torch.tensor(theta, requires_grad=True)
    a = torch.tensor(A, requires_grad=True)
    vt = dec(t, a)
    et = torch.ones_like(vt) if Et is None else torch.tensor(Et)
    g_t, g_a = torch.autograd.grad(vt, (t, a), et, create_graph=True)
    out = {"Vt": vt.detach().numpy(), "E": g_t.detach().numpy(),
           "A_grad_is_A": bool(torch.equal(g_a.detach(), a.detach()))}
    if want_second and Z is not None:
        z = torch.tensor(Z)
        et_leaf = et.clone().requires_grad_()
        # rebuild so that Et is a leaf: Vtd is the grad w.r.t. Et (nw.py:386)
        vt2 = dec(t, a)
        g2, _ = torch.autograd.grad(vt2, (t, a), et_leaf, create_graph=True)
        loss = (g2 * z).sum()
        ed, vtd = torch.autograd.grad(loss, (t, et_leaf), allow_unused=True)
        out["Ed"] = ed.numpy()
        out["Vtd"] = vtd.numpy()
        # second-order grad w.r.t. A is None in the reference (nw.py:386)
        loss2 = (dec.decode(t, a) * z).sum()
        t.grad = None
        a.grad = None
        loss2.backward()
        out["A_second_grad_is_None"] = a.grad is None
        assert np.array_equal(t.grad.numpy(), out["Ed"]) or Et is not None
    return out


def make_known_answer():
    """deepblast/tests/test_nw.py:10-19 (make_data): 5x4 theta = 1/(dist+0.1)."""
    from sklearn.metrics.pairwise import pairwise_distances
    rng = np.random.RandomState(0)
    m, n, k = 2, 1, 3
    Mm = rng.randn(k, 3)
    X = rng.randn(m, 3)
    Y = rng.randn(n, 3)
    X = np.concatenate((X, Mm), axis=0)
    Y = np.concatenate((Mm, Y), axis=0)
    return 1 / (pairwise_distances(X, Y) + 0.1)


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}  {os.path.getsize(path) / 1024:.1f} KiB")


def main():
    os.makedirs(OUT, exist_ok=True)
    meta = {"generated": time.strftime("%Y-%m-%d"), "reference": "flatironinstitute/deepblast @ 2024-11-15",
            "torch": torch.__version__, "numpy": np.__version__, "cases": []}

    # G2: the reference's own known-answer case (float64 input, A = 0.1)
    theta = make_known_answer()[None]
    A = np.ones_like(theta) * 0.1
    for kind in ("nw", "sw"):
        r = run_ref(kind, theta, A, want_second=False)
        dec = DEC[kind]("softmax")
        tb = dec.traceback(torch.tensor(r["E"][0]))
        save(f"g2_known_{kind}.npz", theta=theta, A=A, Vt=r["Vt"], E=r["E"],
             traceback=np.array(tb, dtype=np.int64))
        meta["cases"].append(f"g2_known_{kind}")
    # the same fixture through float32 (what the GPU path sees)
    for kind in ("nw", "sw"):
        th32, A32 = theta.astype(np.float32), A.astype(np.float32)
        r = run_ref(kind, th32, A32, want_second=False)
        tb = DEC[kind]("softmax").traceback(torch.tensor(r["E"][0]))
        save(f"g2_known_{kind}_f32.npz", theta=th32, A=A32, Vt=r["Vt"], E=r["E"],
             traceback=np.array(tb, dtype=np.int64))

    # G1/G4: config-1 plumbing case B=4, N=M=64, float32, with double backward
    theta, A = datagen.theta_A(0, 4, 64, 64)
    Z = datagen.normal(5, (4, 64, 64))
    Et = (1.0 + datagen.uniform(9, (4,))).astype(np.float32)
    for kind in ("nw", "sw"):
        r1 = run_ref(kind, theta, A, None, Z)
        r2 = run_ref(kind, theta, A, Et, Z)
        save(f"g1_{kind}_b4_64.npz", theta=theta, A=A, Z=Z, Et=Et,
             Vt=r1["Vt"], E=r1["E"], Ed=r1["Ed"], Vtd=r1["Vtd"],
             Vt_et=r2["Vt"], E_et=r2["E"], Ed_et=r2["Ed"], Vtd_et=r2["Vtd"],
             A_grad_is_A=r1["A_grad_is_A"], A_second_grad_is_None=r1["A_second_grad_is_None"])
        meta["cases"].append(f"g1_{kind}_b4_64")

    # G3/G4: rectangular and degenerate shapes, random A (positive and negative), non-uniform Et
    shapes = [(1, 1), (1, 7), (7, 1), (2, 2), (3, 5), (37, 101), (101, 37), (64, 65), (65, 64),
              (129, 70)]
    for kind in ("nw", "sw"):
        arrs = {}
        for idx, (N, M) in enumerate(shapes):
            B = 2
            theta, A = datagen.theta_A(100 + idx, B, N, M)
            if idx % 2:  # positive gap scores too (test_nw.py:67-72 uses A = rand)
                A = (-A).astype(np.float32)
            Z = datagen.normal(200 + idx, (B, N, M))
            Et = (0.5 + datagen.uniform(300 + idx, (B,))).astype(np.float32)
            r = run_ref(kind, theta, A, Et, Z)
            pre = f"s{idx}_"
            arrs.update({pre + "theta": theta, pre + "A": A, pre + "Z": Z, pre + "Et": Et,
                         pre + "Vt": r["Vt"], pre + "E": r["E"], pre + "Ed": r["Ed"],
                         pre + "Vtd": r["Vtd"]})
        arrs["shapes"] = np.array(shapes, dtype=np.int64)
        save(f"g3_{kind}_shapes.npz", **arrs)
        meta["cases"].append(f"g3_{kind}_shapes")

    # float64 tensors through the reference (gradcheck path, test_nw.py:56-60)
    theta, A = datagen.theta_A(400, 2, 9, 11, np.float64)
    Z = datagen.normal(401, (2, 9, 11), np.float64)
    for kind in ("nw", "sw"):
        r = run_ref(kind, theta, A, None, Z)
        save(f"g3_{kind}_f64.npz", theta=theta, A=A, Z=Z, Vt=r["Vt"], E=r["E"], Ed=r["Ed"],
             Vtd=r["Vtd"])

    # G6: padded batch with per-item lengths -> reference = per-item sliced B=1 calls
    # (deepblast/alignment.py:165-170)
    B, Nmax, Mmax = 3, 40, 48
    theta, A = datagen.theta_A(500, B, Nmax, Mmax)
    lens = np.array([[40, 48], [17, 33], [29, 5]], dtype=np.int32)
    for kind in ("nw", "sw"):
        Vt = np.zeros(B, np.float32)
        E = np.zeros((B, Nmax, Mmax), np.float32)
        for b in range(B):
            n, m = lens[b]
            r = run_ref(kind, theta[b:b + 1, :n, :m].copy(), A[b:b + 1, :n, :m].copy(),
                        want_second=False)
            Vt[b] = r["Vt"][0]
            E[b, :n, :m] = r["E"][0]
        save(f"g6_{kind}_lens.npz", theta=theta, A=A, lens=lens, Vt=Vt, E=E)
        meta["cases"].append(f"g6_{kind}_lens")

    # G5: large sizes, inputs regenerated from datagen seeds; outputs as checksums + samples
    for (N, seed) in ((512, 600), (1024, 601)):
        B = 2 if N == 512 else 1
        theta, A = datagen.theta_A(seed, B, N, N)
        t0 = time.time()
        r = run_ref("nw", theta, A, want_second=False)
        print(f"  reference nw fwd+bwd B={B} N=M={N}: {time.time() - t0:.1f}s")
        E = r["E"].astype(np.float64)
        w = np.cos(np.arange(N * N, dtype=np.float64).reshape(N, N) * 1e-3)
        save(f"g5_nw_{N}.npz", seed=seed, B=B, N=N, Vt=r["Vt"],
             E_sum=E.sum(axis=(1, 2)), E_wsum=(E * w).sum(axis=(1, 2)),
             E_rows=r["E"][:, ::61, :], E_diag=np.stack([np.diagonal(e) for e in r["E"]]),
             E_max=E.max(axis=(1, 2)))
        meta["cases"].append(f"g5_nw_{N}")
    B, N = 1, 512
    theta, A = datagen.theta_A(602, B, N, N)
    r = run_ref("sw", theta, A, want_second=False)
    E = r["E"].astype(np.float64)
    save("g5_sw_512.npz", seed=602, B=B, N=N, Vt=r["Vt"], E_sum=E.sum(axis=(1, 2)),
         E_rows=r["E"][:, ::61, :], E_diag=np.stack([np.diagonal(e) for e in r["E"]]))
    meta["cases"].append("g5_sw_512")

    # traceback fixtures (host logic; nw.py:401-444, sw.py:328-371): real expected-alignment
    # matrices from the reference, plus random matrices where the reference walk may run
    # off the matrix and raise IndexError (recorded, and mirrored by our host code).
    tbs = {}
    k = 0
    for kind in ("nw", "sw"):
        for idx, (N, M) in enumerate([(5, 4), (8, 8), (3, 9), (9, 3), (1, 1), (1, 4), (4, 1),
                                      (33, 21)]):
            theta, A = datagen.theta_A(700 + idx, 1, N, M)
            theta = theta * 3.0
            r = run_ref(kind, theta, A, want_second=False)
            g = r["E"][0]
            try:
                tb = np.array(DEC[kind]("softmax").traceback(torch.tensor(g)), dtype=np.int64)
                ok = True
            except IndexError:
                tb, ok = np.zeros((0, 3), np.int64), False
            tbs[f"t{k}_grad"], tbs[f"t{k}_states"], tbs[f"t{k}_ok"] = g, tb, ok
            k += 1
    for idx, (N, M) in enumerate([(5, 4), (8, 8), (3, 9), (9, 3), (1, 4), (4, 1)]):
        g = datagen.uniform(800 + idx, (N, M), np.float32)
        try:
            tb = np.array(NeedlemanWunschDecoder("softmax").traceback(torch.tensor(g)),
                          dtype=np.int64)
            ok = True
        except IndexError:
            tb, ok = np.zeros((0, 3), np.int64), False
        tbs[f"t{k}_grad"], tbs[f"t{k}_states"], tbs[f"t{k}_ok"] = g, tb, ok
        k += 1
    dm = np.loadtxt("/root/reference/deepblast/tests/data/dm.txt")
    tb = NeedlemanWunschDecoder("softmax").traceback(torch.tensor(dm))
    tbs[f"t{k}_grad"], tbs[f"t{k}_states"], tbs[f"t{k}_ok"] = dm, np.array(tb, dtype=np.int64), True
    tbs["count"] = k + 1
    save("g8_tracebacks.npz", **tbs)

    with open(os.path.join(OUT, "MANIFEST.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
