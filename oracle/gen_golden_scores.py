#!/usr/bin/env python
"""Golden vectors for SURVEY 8f1: the two lines of the reference that build theta and A
(/root/reference/deepblast/alignment.py:122-123), executed with the real torch ops on CPU in fp32, plus their
gradients w.r.t. the embeddings.  Data only; run in the build container."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import datagen  # noqa: E402


def main():
    out = {}
    shapes = [(3, 37, 53, 24), (2, 130, 129, 70), (1, 40, 72, 512), (1, 1, 1, 1), (2, 5, 300, 33)]
    out["shapes"] = np.array(shapes, dtype=np.int64)
    for idx, (B, N, M, D) in enumerate(shapes):
        sc = 3.0 / np.sqrt(D)   # inner products of a few units: both tails of softplus / logsigmoid are visited
        zx, zy = datagen.normal(500 + idx, (B, N, D)) * sc, datagen.normal(510 + idx, (B, M, D)) * sc * 3
        gx, gy = datagen.normal(520 + idx, (B, N, D)) * sc, datagen.normal(530 + idx, (B, M, D)) * sc * 3
        t = [torch.tensor(a.astype(np.float32), requires_grad=True) for a in (zx, zy, gx, gy)]
        theta = F.softplus(torch.einsum('bid,bjd->bij', t[0], t[1]))      # alignment.py:122
        A = F.logsigmoid(torch.einsum('bid,bjd->bij', t[2], t[3]))        # alignment.py:123
        wt = torch.tensor(datagen.normal(540 + idx, (B, N, M)))
        wa = torch.tensor(datagen.normal(550 + idx, (B, N, M)))
        ((theta * wt).sum() + (A * wa).sum()).backward()
        p = f"s{idx}_"
        for name, a in zip(("zx", "zy", "gx", "gy"), t):
            out[p + name] = a.detach().numpy()
            out[p + "d" + name] = a.grad.numpy()
        out[p + "theta"], out[p + "A"] = theta.detach().numpy(), A.detach().numpy()
        out[p + "wt"], out[p + "wa"] = wt.numpy(), wa.numpy()
        print(shapes[idx], float(theta.max()), float(A.min()))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g11_scores.npz"), **out)


if __name__ == "__main__":
    main()
