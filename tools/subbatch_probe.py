"""does the HBM round trip of theta / A between the scores GEMM and the DP go away if the batch is cut into
sub-batches small enough for the Infinity Cache (256 MiB), with scores(k+1) on a second stream under DP(k)?

Orders compared at B=256, N=M=D=512 (scores 0.85 ms, DP fwd+bwd 0.38 ms as separate full-batch launches):
  monolithic      : scores(all) ; fwd(all) ; bwd(all)                              -- what bench.py --mode scores+dp times
  serial sub      : for k: scores(k) ; fwd(k) ; bwd(k)          one stream         -- cache effect alone
  overlapped sub  : for k: [scores(k+1) on stream 2] || [fwd(k) ; bwd(k)]          -- cache effect + overlap
Also prints the reference's own ops for the scores on this box (alignment.py:122-123).
usage: python tools/subbatch_probe.py [sub-batch sizes ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

import datagen  # noqa: E402
from deepblast_amd._engine import get_engine  # noqa: E402
from deepblast_amd.scores import alignment_scores  # noqa: E402

B, N, M, D = 256, 512, 512, 512
subs = [int(a) for a in sys.argv[1:]] or [32, 64, 96, 128]
eng = get_engine()
sc = 2.0 / np.sqrt(D)
emb = [torch.from_numpy((datagen.normal(20 + i, (B, n, D)) * sc).astype(np.float32)).cuda() for i, n in enumerate((N, M, N, M))]
et = torch.ones(B, device="cuda")


def timeit(fn, n=6):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n


def dp(th, a, lo, hi):
    Vt, Q = eng.forward(th, a, 0)
    return eng.backward(et[lo:hi], Q, tuple(th.shape), 0)


def mono():
    th, a = alignment_scores(*emb)
    return dp(th, a, 0, B)


def serial(S):
    def run():
        for lo in range(0, B, S):
            th, a = alignment_scores(*[t[lo:lo + S] for t in emb])
            dp(th, a, lo, lo + S)
    return run


s2 = torch.cuda.Stream()


def overlapped(S):
    def run():
        main = torch.cuda.current_stream()
        nxt = None
        for lo in range(0, B, S):
            if nxt is None:
                th, a = alignment_scores(*[t[lo:lo + S] for t in emb])
            else:
                main.wait_event(nxt[2])
                th, a = nxt[0], nxt[1]
            if lo + S < B:
                s2.wait_stream(main) if lo == 0 else None
                with torch.cuda.stream(s2):
                    t2, a2 = alignment_scores(*[t[lo + S:lo + 2 * S] for t in emb])
                    ev = torch.cuda.Event()
                    ev.record(s2)
                t2.record_stream(main), a2.record_stream(main)
                nxt = (t2, a2, ev)
            dp(th, a, lo, lo + S)
    return run


def torch_scores():
    th = F.softplus(torch.einsum('bid,bjd->bij', emb[0], emb[1]))
    a = F.logsigmoid(torch.einsum('bid,bjd->bij', emb[2], emb[3]))
    return th, a


print(f"B={B} N={N} M={M} D={D}")
print(f"scores alone (sdp_scores_f32)           : {timeit(lambda: alignment_scores(*emb)):.3f} ms")
print(f"scores alone (torch einsum + activation): {timeit(torch_scores):.3f} ms   <- the reference's own ops, alignment.py:122-123")
th, a = alignment_scores(*emb)
print(f"DP fwd+bwd alone                        : {timeit(lambda: dp(th, a, 0, B)):.3f} ms")
del th, a
print(f"monolithic scores ; fwd ; bwd           : {timeit(mono):.3f} ms")
for S in subs:
    print(f"sub-batches of {S:3d}: serial {timeit(serial(S)):.3f} ms   overlapped (2 streams) {timeit(overlapped(S)):.3f} ms", flush=True)
