#!/usr/bin/env python
"""Golden vectors for SURVEY 8f3: run the REAL reference losses (/root/reference/deepblast/losses.py:
MatrixCrossEntropy, SoftPathLoss, SoftAlignmentLoss) on small inputs and store inputs, loss values and
the gradients w.r.t. the predicted alignment matrix.  Data only; run in the build container."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(1, "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tests"))
from deepblast.losses import MatrixCrossEntropy, SoftAlignmentLoss, SoftPathLoss  # noqa: E402
import datagen  # noqa: E402
from oracle import oracle  # noqa: E402  (only to make realistic predicted matrices)

sys.path.insert(0, ROOT)


def main():
    B, N, M = 5, 40, 56
    lens = np.array([[40, 56], [17, 33], [29, 5], [1, 1], [40, 1]], dtype=np.int64)
    theta, A = datagen.theta_A(900, B, N, M)
    _, E, _, _ = oracle.fwd_bwd(theta * 2, A, None, 0)
    Yp = E.astype(np.float32)
    Yp[0, 0, 0] = 0.0          # exercises the clamp in MatrixCrossEntropy (losses.py:27-28)
    Yp[1, 3, 3] = 1.0
    Yt = (datagen.uniform(901, (B, N, M)) < 0.08).astype(np.float32)
    P = (datagen.uniform(902, (B, N, M)) * 5).astype(np.float32)
    G = (datagen.uniform(903, (B, N, M)) < 0.7).astype(np.float32)
    G[3] = 1.0
    out = {"Yt": Yt, "Yp": Yp, "P": P, "G": G, "lens": lens}
    xl, yl = lens[:, 0].tolist(), lens[:, 1].tolist()
    for name, fn, first in (("mce", MatrixCrossEntropy(), Yt), ("path", SoftPathLoss(), P),
                            ("align", SoftAlignmentLoss(), Yt)):
        yp = torch.tensor(Yp, requires_grad=True)
        loss = fn(torch.tensor(first), yp, xl, yl, torch.tensor(G))
        loss.backward()
        out[name + "_loss"] = loss.detach().numpy()
        out[name + "_grad"] = yp.grad.numpy()
        print(name, float(loss))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g9_losses.npz"), **out)


if __name__ == "__main__":
    main()
