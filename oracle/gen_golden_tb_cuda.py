#!/usr/bin/env python
"""tests/golden/g12_tracebacks_cuda.npz: what the traceback of the reference's GPU decoder classes
(/root/reference/deepblast/nw_cuda.py:273-317, sw_cuda.py:283-327) returns -- the classes deepblast_amd replaces.
Their walk differs from the CPU classes' (nw.py:401-444): it stops as soon as ANY neighbour is off the matrix.

The modules import under oracle/_shim/numba (their @cuda.jit kernels are never called); traceback() is host code and
takes a CPU tensor.  Matrices: the g8 set (real alignment matrices and random ones), the reference's own dm.txt, and
matrices built to reach row 0 / column 0 early.  Run in the build container:  python oracle/gen_golden_tb_cuda.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "_shim"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tests"))

from deepblast.nw_cuda import NeedlemanWunschDecoder as NWCuda  # noqa: E402
from deepblast.sw_cuda import SmithWatermanDecoder as SWCuda  # noqa: E402
import datagen  # noqa: E402


def main():
    g8 = np.load(os.path.join(ROOT, "tests", "golden", "g8_tracebacks.npz"))
    mats = [g8[f"t{k}_grad"] for k in range(int(g8["count"]))]
    # walks that reach an edge early: mass along the top row / left column, a band near a corner, flat matrices
    for idx, (N, M) in enumerate([(6, 9), (9, 6), (12, 12), (1, 7), (7, 1), (2, 2), (40, 25), (64, 64), (70, 130)]):
        g = datagen.uniform(900 + idx, (N, M), np.float32)
        if idx % 3 == 0:
            g[0, :] += 2.0
        if idx % 3 == 1:
            g[:, 0] += 2.0
        mats.append(g)
    mats.append(np.zeros((5, 8), np.float32))
    mats.append(np.full((4, 4), -1e10, np.float32))   # every neighbour equals the sentinel: stops at once
    out = {"count": len(mats)}
    for k, g in enumerate(mats):
        for name, dec in (("nw", NWCuda("softmax")), ("sw", SWCuda("softmax"))):
            tb = dec.traceback(torch.tensor(np.asarray(g)))
            out[f"t{k}_{name}"] = np.array(tb, dtype=np.int64)
        out[f"t{k}_grad"] = np.asarray(g)
        assert np.array_equal(out[f"t{k}_nw"], out[f"t{k}_sw"])   # the two classes' walks are the same code
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g12_tracebacks_cuda.npz"), **out)
    print("wrote g12_tracebacks_cuda.npz:", len(mats), "matrices")


if __name__ == "__main__":
    main()
