#!/usr/bin/env python
"""Timings of BASELINE.json configs[1..3] on one GPU through the public API (for DESIGN.md)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import datagen  # noqa: E402
from deepblast_amd import NeedlemanWunschDecoder, SmithWatermanDecoder  # noqa: E402


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def step(dec, theta, A, lens=None, fill=True):
    t = theta.detach().requires_grad_(True)
    v = (dec(t, A, lens) if fill else dec(t, A, lens, fill=False)) if lens is not None else dec(t, A)
    v.sum().backward()


def main():
    B = 256
    th, A = datagen.theta_A(1, B, 512, 512)
    th, A = torch.from_numpy(th).cuda(), torch.from_numpy(A).cuda()
    for name, Dec in (("configs[1] NW 512x512", NeedlemanWunschDecoder), ("configs[3] SW 512x512", SmithWatermanDecoder)):
        ms = timeit(lambda: step(Dec("softmax"), th, A))
        cells = B * 512 * 512 if "NW" in name else B * 511 * 511
        print(f"{name}: {ms:.3f} ms/step  {2 * cells / ms * 1e3:.3e} cell-updates/s")
    lens = datagen.lengths(2, B, 64, 1024)
    N, M = int(lens[:, 0].max()), int(lens[:, 1].max())
    th, A = datagen.theta_A(2, B, N, M)
    th, A = torch.from_numpy(th).cuda(), torch.from_numpy(A).cuda()
    ln = torch.from_numpy(lens).cuda()
    dec = NeedlemanWunschDecoder("softmax")
    work = int((lens[:, 0].astype(np.int64) * lens[:, 1]).sum())
    ms = timeit(lambda: step(dec, th, A), 5)
    print(f"configs[2] padded ({B},{N},{M}) reference semantics: {ms:.3f} ms/step  {2 * B * N * M / ms * 1e3:.3e} padded cell-updates/s"
          f"  ({2 * work / ms * 1e3:.3e} counting true lengths)")
    ms = timeit(lambda: step(dec, th, A, ln), 5)
    print(f"configs[2] lengths-aware: {ms:.3f} ms/step  {2 * work / ms * 1e3:.3e} true cell-updates/s (sum n_b*m_b = {work})")
    # what a consumer that masks by the same lengths runs (decode_loss, traceback_batch(E, lengths), gather="paths"): E outside
    # the pairs' blocks is not zero-filled (include/sdp.h: SDP_NO_FILL; dec(theta, A, lengths, fill=False))
    ms = timeit(lambda: step(dec, th, A, ln, fill=False), 5)
    print(f"configs[2] lengths-aware, no zero fill outside the blocks (SDP_NO_FILL): {ms:.3f} ms/step  {2 * work / ms * 1e3:.3e} true cell-updates/s")
    from deepblast_amd._engine import get_engine
    eng = get_engine()
    Vt, Q = eng.forward(th, A, 0, ln)
    et = torch.ones(B, device="cuda")
    for nf in (False, True):
        ms = timeit(lambda: eng.backward(et, Q, (B, N, M), 0, ln, no_fill=nf), 5)
        print(f"   backward sweep alone, lengths-aware, no_fill={nf}: {ms * 1e3:.1f} us")
    ms = timeit(lambda: eng.forward(th, A, 0, ln), 5)
    print(f"   forward sweep alone, lengths-aware: {ms * 1e3:.1f} us")


if __name__ == "__main__":
    main()
