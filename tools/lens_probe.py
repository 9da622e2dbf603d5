"""configs[2] (256 pairs, n, m ~ U[64,1024], padded) lengths-aware: where does the time go?  Per-kernel times for the
batch, for the batch sorted by work, with forced wave counts, and for its largest pair alone."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import gpu_tune
sys.path.insert(0, os.path.join(gpu_tune.ROOT, "tests"))
import datagen
from deepblast_amd._engine import get_engine
eng = get_engine()
B = 256
lens = datagen.lengths(2, B, 64, 1024)
N, M = int(lens[:, 0].max()), int(lens[:, 1].max())
if "align" in sys.argv:   # pad the column count to a multiple of 32: rows start on 128-byte lines
    M = (M + 31) // 32 * 32
print("padded shape", (B, N, M), flush=True)
th, A = datagen.theta_A(2, B, N, M)
th, A = torch.from_numpy(th).cuda(), torch.from_numpy(A).cuda()
et = torch.ones(B, device="cuda")


def run(tag, theta, gap, ln, waves=0):
    eng.force_waves = {0: waves, 1: waves} if waves else {}
    Bn = theta.shape[0]
    lnt = None if ln is None else torch.from_numpy(np.ascontiguousarray(ln)).cuda()
    st = {}
    def f():
        st["vt"], st["q"] = eng.forward(theta, gap, 0, lnt)
    def b():
        st["e"] = eng.backward(et[:Bn], st["q"], tuple(theta.shape), 0, lnt)
    f()
    tf = gpu_tune.timeit(f, 5)
    tb = gpu_tune.timeit(b, 5)
    work = int((ln[:, 0].astype(np.int64) * ln[:, 1]).sum()) if ln is not None else Bn * theta.shape[1] * theta.shape[2]
    print(f"{tag:34s} fwd={tf:8.1f} bwd={tb:8.1f} us   {2 * work / (tf + tb) * 1e6:.3e} true cell-updates/s", flush=True)
    eng.force_waves = {}


run("batch order, auto", th, A, lens)
for w in (2, 4, 8):
    run(f"batch order, W={w}", th, A, lens, w)
order = np.argsort(-(lens[:, 0].astype(np.int64) * lens[:, 1]))
run("sorted by work desc, auto", th[order].contiguous(), A[order].contiguous(), lens[order])
big = int(order[0])
run(f"largest pair alone {tuple(lens[big])}", th[big:big + 1].contiguous(), A[big:big + 1].contiguous(), lens[big:big + 1])
for w in (4, 8):
    run(f"largest pair alone W={w}", th[big:big + 1].contiguous(), A[big:big + 1].contiguous(), lens[big:big + 1], w)
run("top-16 pairs", th[order[:16]].contiguous(), A[order[:16]].contiguous(), lens[order[:16]])
run("full padded (no lens)", th, A, None)
