#!/usr/bin/env python
"""(test infrastructure) Diagnostic sweep on a GPU box: selftest, parity table over many shapes, quick timing.
Usage: python tests/gpu_check.py [--quick] [--time]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # parity.py / datagen.py live here

import torch  # noqa: E402

import datagen  # noqa: E402
import parity  # noqa: E402
from deepblast_amd._engine import get_engine  # noqa: E402


def main():
    quick = "--quick" in sys.argv
    eng = get_engine()
    print("device:", torch.cuda.get_device_name(0), flush=True)
    try:
        eng.selftest(0)
        print("selftest: ok", flush=True)
    except Exception as e:  # keep going: the parity table says more
        print("selftest: FAILED", e, flush=True)

    shapes = [(1, 1, 1), (2, 1, 7), (2, 7, 1), (2, 2, 2), (3, 5, 4), (2, 16, 16), (2, 17, 33), (2, 37, 101),
              (2, 64, 64), (2, 63, 65), (2, 65, 63), (2, 65, 64), (2, 101, 37), (2, 128, 128), (2, 129, 70),
              (2, 200, 300), (3, 257, 255), (2, 320, 90), (4, 512, 512)]
    if not quick:
        shapes += [(2, 700, 1024), (1, 1024, 1024), (1, 70, 2048)]
    worst = 0.0
    nfail = 0
    for variant in (0, 1):
        for idx, (B, N, M) in enumerate(shapes):
            theta, A = datagen.theta_A(1000 + idx, B, N, M)
            if idx % 3 == 1:
                A = (-A).astype(np.float32)
            Z = datagen.normal(2000 + idx, (B, N, M))
            Et = (0.5 + datagen.uniform(3000 + idx, (B,))).astype(np.float32)
            t0 = time.time()
            ref = parity.oracle_all(theta, A, Et, Z, variant)
            t1 = time.time()
            try:
                got = parity.engine_all(theta, A, Et, Z, variant)
                errs = parity.compare(got, ref)
            except Exception as e:
                print(f"v{variant} {B}x{N}x{M}: EXCEPTION {e}", flush=True)
                nfail += 1
                continue
            bad = max(errs.values())
            bad = bad if np.isfinite(bad) else 9e9
            worst = max(worst, bad)
            flag = "" if bad <= parity.TOL else "  <-- FAIL"
            nfail += bad > parity.TOL
            print(f"v{variant} {B}x{N}x{M}: " + " ".join(f"{k}={v:.2e}" for k, v in errs.items())
                  + f"  (oracle {t1 - t0:.2f}s){flag}", flush=True)
    # lengths-aware
    for variant in (0, 1):
        B, N, M = 6, 150, 170
        theta, A = datagen.theta_A(4000, B, N, M)
        Z = datagen.normal(4001, (B, N, M))
        lens = datagen.lengths(4002, B, 1, 150)
        lens[0] = (N, M)
        ref = parity.oracle_lens(theta, A, None, Z, variant, lens)
        try:
            got = parity.engine_all(theta, A, None, Z, variant, lens=lens)
            errs = parity.compare(got, ref)
            bad = max(errs.values())
            bad = bad if np.isfinite(bad) else 9e9
        except Exception as e:
            print(f"v{variant} lens: EXCEPTION {e}")
            errs, bad = {}, 9e9
        nfail += bad > parity.TOL
        print(f"v{variant} lens {B}x{N}x{M}: " + " ".join(f"{k}={v:.2e}" for k, v in errs.items())
              + ("" if bad <= parity.TOL else "  <-- FAIL"), flush=True)

    # extreme magnitudes (large match scores, very negative / positive gap scores)
    for variant in (0, 1):
        for name, ts, asc, ash in (("theta*40", 40.0, 1.0, 0.0), ("A*150", 1.0, 150.0, 0.0), ("theta*90,A*300", 90.0, 300.0, 0.0),
                                   ("A+5", 1.0, 1.0, 5.0), ("theta*1e-3", 1e-3, 1e-3, 0.0)):
            B, N, M = 2, 130, 97
            theta, A = datagen.theta_A(5000, B, N, M)
            theta = (theta * ts).astype(np.float32)
            A = (A * asc + ash).astype(np.float32)
            Z = datagen.normal(5001, (B, N, M))
            ref = parity.oracle_all(theta, A, None, Z, variant)
            got = parity.engine_all(theta, A, None, Z, variant)
            errs = parity.compare(got, ref)
            bad = max(errs.values())
            bad = bad if np.isfinite(bad) else 9e9
            worst = max(worst, bad)
            nfail += bad > parity.TOL
            print(f"v{variant} stress {name}: " + " ".join(f"{k}={v:.2e}" for k, v in errs.items())
                  + ("" if bad <= parity.TOL else "  <-- FAIL"), flush=True)

    if "--fuzz" in sys.argv:
        nf = int(sys.argv[sys.argv.index("--fuzz") + 1])
        rng = np.random.default_rng(12345)
        fworst = 0.0
        for it in range(nf):
            B = int(rng.integers(1, 5))
            N = int(rng.choice([rng.integers(1, 40), rng.integers(40, 200), rng.integers(200, 700)]))
            M = int(rng.choice([rng.integers(1, 40), rng.integers(40, 200), rng.integers(200, 900)]))
            variant = int(rng.integers(0, 2))
            theta, A = datagen.theta_A(10000 + it, B, N, M)
            theta = (theta * float(rng.choice([0.1, 1.0, 5.0]))).astype(np.float32)
            A = (A * float(rng.choice([0.1, 1.0, 10.0])) + float(rng.choice([0.0, 0.0, 0.5]))).astype(np.float32)
            Z = datagen.normal(20000 + it, (B, N, M))
            use_lens = bool(rng.integers(0, 2))
            try:
                if use_lens:
                    lens = np.stack([rng.integers(1, N + 1, B), rng.integers(1, M + 1, B)], axis=1).astype(np.int32)
                    ref = parity.oracle_lens(theta, A, None, Z, variant, lens)
                    got = parity.engine_all(theta, A, None, Z, variant, lens=lens)
                else:
                    Et = (0.5 + datagen.uniform(30000 + it, (B,))).astype(np.float32)
                    ref = parity.oracle_all(theta, A, Et, Z, variant)
                    got = parity.engine_all(theta, A, Et, Z, variant)
                errs = parity.compare(got, ref)
                bad = max(errs.values())
                bad = bad if np.isfinite(bad) else 9e9
            except Exception as e:
                print("fuzz EXCEPTION", it, B, N, M, variant, use_lens, e, flush=True)
                bad = 9e9
            fworst = max(fworst, bad)
            if bad > parity.TOL:
                nfail += 1
                print(f"fuzz FAIL it={it} B={B} N={N} M={M} v={variant} lens={use_lens}: {errs}", flush=True)
        worst = max(worst, fworst)
        print(f"fuzz: {nf} random cases, worst {fworst:.3e}", flush=True)

    print(f"worst normalised error {worst:.3e}; failures {nfail}", flush=True)

    if "--time" in sys.argv:
        for (B, N, M) in ((256, 512, 512),):
            theta, A = datagen.theta_A(1, B, N, M)
            t = torch.from_numpy(theta).cuda()
            a = torch.from_numpy(A).cuda()
            et = torch.ones(B, device="cuda")
            z = torch.from_numpy(datagen.normal(7, (B, N, M))).cuda()
            for name, fn in (("fwd", lambda: eng.forward(t, a, 0)),):
                pass
            Vt, Q = eng.forward(t, a, 0)
            E = eng.backward(et, Q, (B, N, M), 0)
            torch.cuda.synchronize()

            def timeit(fn, n=10):
                fn()
                torch.cuda.synchronize()
                s = torch.cuda.Event(enable_timing=True)
                e = torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(n):
                    fn()
                e.record()
                torch.cuda.synchronize()
                return s.elapsed_time(e) / n

            tf = timeit(lambda: eng.forward(t, a, 0))
            tb = timeit(lambda: eng.backward(et, Q, (B, N, M), 0))
            _, Qx = eng.forward(t, a, 0, exact_state=True)
            Vtd, Qd = eng.adjoint_forward(Qx, z, None, 0)
            taf = timeit(lambda: eng.adjoint_forward(Qx, z, None, 0))
            tab = timeit(lambda: eng.adjoint_backward(E, Qx, Qd, 0))
            cells = B * N * M
            print(f"timing B={B} N={N} M={M}: fwd {tf:.3f} ms  bwd {tb:.3f} ms  adj_fwd {taf:.3f} ms  adj_bwd {tab:.3f} ms")
            print(f"  fwd+bwd {tf + tb:.3f} ms -> {2 * cells / ((tf + tb) * 1e-3):.3e} cell-updates/s ; "
                  f"roofline frac (24 B/cell @ 8 TB/s) {(cells * 24 / ((tf + tb) * 1e-3)) / 8e12:.3f}")
    return 1 if nfail else 0


if __name__ == "__main__":
    sys.exit(main())
