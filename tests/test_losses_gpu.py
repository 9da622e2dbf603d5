"""GPU: fused masked losses against fixtures produced by the real reference
(deepblast/losses.py, oracle/gen_golden_losses.py) and against a plain-torch fp32 restatement at the
headline size.  Tolerance: 1e-5 relative on the scalar, 1e-5 * max|grad| absolute on the gradient
(the reference accumulates in fp32; the kernel in float64)."""
import os
import time

import numpy as np
import pytest
import torch

import datagen
from deepblast_amd.losses import MatrixCrossEntropy, SoftAlignmentLoss, SoftPathLoss

pytestmark = pytest.mark.gpu
LOSS = {"mce": (MatrixCrossEntropy, "Yt"), "path": (SoftPathLoss, "P"), "align": (SoftAlignmentLoss, "Yt")}


def _torch_reference(name, first, pred, xl, yl, G):
    """The reference algorithm restated with the same torch ops (losses.py:26-46, 69-79, 108-118)."""
    score = 0
    if name == "mce":
        eps = 3e-8
        pred = torch.clamp(pred, min=eps, max=1 - eps)
    for b in range(len(xl)):
        sl = (b, slice(0, xl[b]), slice(0, yl[b]))
        g = G[sl].bool()
        if name == "mce":
            v = first[sl] * torch.log(pred[sl]) + (1 - first[sl]) * torch.log(1 - pred[sl])
            score = score - torch.mean(torch.masked_select(v, g))
        elif name == "path":
            score = score + torch.norm(torch.masked_select(first[sl] * pred[sl], g))
        else:
            score = score + torch.norm(torch.masked_select(first[sl] - pred[sl], g))
    return score / len(xl)


@pytest.mark.parametrize("name", ["mce", "path", "align"])
def test_against_reference_fixture(golden_dir, name):
    d = np.load(os.path.join(golden_dir, "g9_losses.npz"))
    cls, first = LOSS[name]
    pred = torch.from_numpy(d["Yp"]).cuda().requires_grad_()
    lens = d["lens"]
    loss = cls()(torch.from_numpy(d[first]).cuda(), pred, lens[:, 0].tolist(), lens[:, 1].tolist(),
                 torch.from_numpy(d["G"]).cuda())
    loss.backward()
    assert abs(float(loss) - float(d[name + "_loss"])) <= 1e-5 * max(1.0, abs(float(d[name + "_loss"])))
    gref = d[name + "_grad"]
    assert np.max(np.abs(pred.grad.cpu().numpy() - gref)) <= 1e-5 * max(1.0, np.abs(gref).max())


@pytest.mark.parametrize("name", ["mce", "path", "align"])
def test_headline_size_and_timing(name):
    B, N, M = 256, 512, 512
    lens = datagen.lengths(70, B, 64, 512)
    Yp = torch.from_numpy(datagen.uniform(71, (B, N, M)) * 0.98 + 0.01).cuda()
    Yt = torch.from_numpy((datagen.uniform(72, (B, N, M)) < 0.05).astype(np.float32)).cuda()
    P = torch.from_numpy(datagen.uniform(73, (B, N, M)) * 4).cuda()
    G = torch.from_numpy((datagen.uniform(74, (B, N, M)) < 0.8).astype(np.float32)).cuda()
    first = P if name == "path" else Yt
    xl, yl = lens[:, 0].tolist(), lens[:, 1].tolist()
    cls = LOSS[name][0]()

    def ours():
        p = Yp.detach().requires_grad_()
        loss = cls(first, p, xl, yl, G)
        loss.backward()
        return loss.detach(), p.grad

    def ref():
        p = Yp.detach().requires_grad_()
        loss = _torch_reference(name, first, p, xl, yl, G)
        loss.backward()
        return loss.detach(), p.grad

    l1, g1 = ours()
    l2, g2 = ref()
    assert abs(float(l1) - float(l2)) <= 2e-5 * max(1.0, abs(float(l2)))
    assert float((g1 - g2).abs().max()) <= 2e-5 * max(1.0, float(g2.abs().max()))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        ours()
    torch.cuda.synchronize()
    t_ours = (time.perf_counter() - t0) / 5
    t0 = time.perf_counter()
    ref()
    torch.cuda.synchronize()
    t_ref = time.perf_counter() - t0
    print(f"loss {name} B={B} {N}x{M}: fused {t_ours * 1e3:.2f} ms fwd+bwd, per-pair torch loop {t_ref * 1e3:.1f} ms")
