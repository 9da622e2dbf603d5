"""GPU: the widened path end to end (SURVEY 8 f4 -> a -> f3): reference-style collate with the lengths side-channel
-> decode(theta, A, lengths) -> fused masked loss with the same lengths -> backward.  Expected values: the reference's
per-item procedure (alignment.py:165-170 slices each pair; losses.py:26-46 loops over pairs) evaluated with the CPU
oracle in float64."""
import os

import numpy as np
import pytest
import torch

import datagen
import parity
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("variant", [0, 1], ids=["nw", "sw"])
def test_collate_decode_loss_backward(golden_dir, variant):
    from deepblast_amd import NeedlemanWunschDecoder, SmithWatermanDecoder
    from deepblast_amd.batching import collate_with_lengths
    from deepblast_amd.losses import MatrixCrossEntropy
    d = np.load(os.path.join(golden_dir, "g10_batching.npz"))
    batch = [tuple(torch.from_numpy(d[f"i{b}_{k}"]) for k in ("gene", "other", "states", "aln", "path", "mask", "gm", "om"))
             for b in range(len(d["sizes"]))]
    genes, others, states, dm, p, G, gM, oM, lengths = collate_with_lengths(batch)
    B, N, M = dm.shape
    G = G.clone()
    G[:, 0, 0] = True   # every pair counts at least one cell (an empty mask is mean([]) = NaN in the reference too)
    theta, A = datagen.theta_A(333, B, N, M)   # stands in for the language-model scores (alignment.py:122-123)
    Yt = (dm > 0.8).float()
    dev = torch.device("cuda", 0)
    t = torch.from_numpy(theta).to(dev).requires_grad_()
    a = torch.from_numpy(A).to(dev).requires_grad_()
    dec = (NeedlemanWunschDecoder, SmithWatermanDecoder)[variant]("softmax")
    aln = dec.decode(t, a, lengths.to(dev))
    xl, yl = lengths[:, 0].tolist(), lengths[:, 1].tolist()
    loss = MatrixCrossEntropy()(Yt.to(dev), aln, xl, yl, G.to(dev))
    loss.backward()
    torch.cuda.synchronize()

    # reference procedure, pair by pair
    ref_loss, ref_grad, ref_aln = 0.0, np.zeros((B, N, M)), np.zeros((B, N, M), np.float32)
    eps = 3e-8
    for b in range(B):
        n, m = xl[b], yl[b]
        th, ga = np.ascontiguousarray(theta[b:b + 1, :n, :m]), np.ascontiguousarray(A[b:b + 1, :n, :m])
        _, E, Q, Ef = oracle.fwd_bwd(th, ga, None, variant)
        ref_aln[b, :n, :m] = E[0]
        g = G[b, :n, :m].numpy().astype(bool)
        y = Yt[b, :n, :m].numpy().astype(np.float64)
        pr = np.clip(E[0].astype(np.float32), np.float32(eps), np.float32(1 - eps)).astype(np.float64)
        cnt = g.sum()
        ref_loss += -np.sum((y * np.log(pr) + (1 - y) * np.log(1 - pr))[g]) / cnt / B
        inside = (E[0] >= eps) & (E[0] <= 1 - eps)
        Z = np.where(g & inside, -(y / pr - (1 - y) / (1 - pr)) / (cnt * B), 0.0).astype(np.float32)
        Ed, _, _ = oracle.double_backward(Q, Ef, Z[None])
        ref_grad[b, :n, :m] = Ed[0]
    assert parity.abs_err(aln.detach().cpu().numpy(), ref_aln) <= parity.TOL
    assert abs(float(loss) - ref_loss) <= 1e-4 * max(1.0, abs(ref_loss))
    assert parity.abs_err(t.grad.cpu().numpy(), ref_grad, scale=True) <= parity.TOL
    # nothing leaks outside a pair's own block
    for b in range(B):
        assert not t.grad[b, xl[b]:, :].any() and not t.grad[b, :, yl[b]:].any()


@pytest.mark.parametrize("name", ["mce", "path", "align"])
@pytest.mark.parametrize("use_lengths", [False, True], ids=["padded-dp", "lengths-dp"])
def test_fused_decode_loss_equals_the_unfused_path(name, use_lengths):
    """SURVEY f3 as written: the masked loss's gradient seeds the adjoint forward sweep inside the kernel
    (sdp_adjoint_forward_loss_f32).  Same loss and the same gradient w.r.t. theta as decode() -> loss -> backward."""
    from deepblast_amd import NeedlemanWunschDecoder
    from deepblast_amd.losses import MatrixCrossEntropy, SoftAlignmentLoss, SoftPathLoss, decode_loss
    B, N, M = 5, 150, 170
    theta, A = datagen.theta_A(1234, B, N, M)
    lens = datagen.lengths(1235, B, 20, 150)
    lens[0] = (N, M)
    dev = torch.device("cuda", 0)
    Yt = torch.from_numpy((datagen.uniform(1236, (B, N, M)) < 0.1).astype(np.float32)).to(dev)
    P = torch.from_numpy(datagen.uniform(1237, (B, N, M)) * 3).to(dev)
    G = torch.from_numpy((datagen.uniform(1238, (B, N, M)) < 0.8).astype(np.float32)).to(dev)
    loss_fn, first = {"mce": (MatrixCrossEntropy(), Yt), "path": (SoftPathLoss(), P), "align": (SoftAlignmentLoss(), Yt)}[name]
    xl, yl = lens[:, 0].tolist(), lens[:, 1].tolist()
    ln = torch.from_numpy(lens).to(dev) if use_lengths else None
    dec = NeedlemanWunschDecoder("softmax")

    t1 = torch.from_numpy(theta).to(dev).requires_grad_()
    a1 = torch.from_numpy(A).to(dev).requires_grad_()
    aln = dec.decode(t1, a1, ln) if use_lengths else dec.decode(t1, a1)
    l1 = loss_fn(first, aln, xl, yl, G)
    l1.backward()

    t2 = torch.from_numpy(theta).to(dev).requires_grad_()
    a2 = torch.from_numpy(A).to(dev).requires_grad_()
    l2, E = decode_loss(dec, loss_fn, t2, a2, first, xl, yl, G, lengths=ln)
    l2.backward()
    torch.cuda.synchronize()
    if use_lengths:
        # decode_loss leaves E outside the pairs' blocks unwritten by default (SDP_NO_FILL: nothing in the op reads it); inside
        # the blocks it is the decoder's E bit for bit, and fill=True gives the whole tensor
        for b in range(B):
            assert torch.equal(E[b, :xl[b], :yl[b]], aln.detach()[b, :xl[b], :yl[b]]), b
        t3 = torch.from_numpy(theta).to(dev).requires_grad_()
        l3, E3 = decode_loss(dec, loss_fn, t3, torch.from_numpy(A).to(dev), first, xl, yl, G, lengths=ln, fill=True)
        assert torch.equal(E3, aln.detach()) and float(l3) == float(l2)
    else:
        assert torch.equal(E, aln.detach())
    assert abs(float(l1) - float(l2)) <= 1e-6 * max(1.0, abs(float(l1)))
    g1, g2 = t1.grad, t2.grad
    assert float((g1 - g2).abs().max()) <= 1e-6 * max(1.0, float(g1.abs().max())), float((g1 - g2).abs().max())
    assert a2.grad is None


@pytest.mark.parametrize("name", ["mce", "path", "align"])
def test_fused_decode_loss_when_the_loss_reads_beyond_the_dp_blocks(name):
    """decode_loss(lengths=...) with x_len / y_len LARGER than `lengths` -- the API allows the loss to slice by other
    lengths than the DP sweeps.  The loss kernel then reads E between the two blocks, which decode() zero-fills; the fused op
    must not leave it unwritten there (round 5 did, by default: garbage or NaN in the loss).  Same loss and gradient as the
    unfused path, with the buffers poisoned beforehand so that unwritten memory cannot pass for zeros."""
    from deepblast_amd import NeedlemanWunschDecoder
    from deepblast_amd.losses import MatrixCrossEntropy, SoftAlignmentLoss, SoftPathLoss, decode_loss
    B, N, M = 6, 140, 150
    theta, A = datagen.theta_A(4321, B, N, M)
    lens = datagen.lengths(4322, B, 20, 100)
    big = np.minimum(lens + np.array([[30, 40]]), np.array([[N, M]])).astype(lens.dtype)   # the loss's lengths: beyond the DP's
    dev = torch.device("cuda", 0)
    Yt = torch.from_numpy((datagen.uniform(4323, (B, N, M)) < 0.1).astype(np.float32)).to(dev)
    P = torch.from_numpy(datagen.uniform(4324, (B, N, M)) * 3).to(dev)
    G = torch.from_numpy((datagen.uniform(4325, (B, N, M)) < 0.8).astype(np.float32)).to(dev)
    loss_fn, first = {"mce": (MatrixCrossEntropy(), Yt), "path": (SoftPathLoss(), P), "align": (SoftAlignmentLoss(), Yt)}[name]
    xl, yl = big[:, 0].tolist(), big[:, 1].tolist()
    dec = NeedlemanWunschDecoder("softmax")
    t1 = torch.from_numpy(theta).to(dev).requires_grad_()
    a1 = torch.from_numpy(A).to(dev).requires_grad_()   # (decode differentiates w.r.t. both, like the reference)
    l1 = loss_fn(first, dec.decode(t1, a1, lens.tolist()), xl, yl, G)
    l1.backward()
    # poison the allocator's free blocks: whatever the fused op leaves unwritten is then NaN, not a lucky zero
    for _ in range(3):
        junk = torch.full((B, N, M), float("nan"), device=dev)
        del junk
    t2 = torch.from_numpy(theta).to(dev).requires_grad_()
    l2, E = decode_loss(dec, loss_fn, t2, torch.from_numpy(A).to(dev).requires_grad_(), first, xl, yl, G, lengths=lens.tolist())
    l2.backward()
    torch.cuda.synchronize()
    assert torch.isfinite(l2) and torch.isfinite(E).all()
    assert abs(float(l1) - float(l2)) <= 1e-6 * max(1.0, abs(float(l1))), (float(l1), float(l2))
    assert float((t1.grad - t2.grad).abs().max()) <= 1e-6 * max(1.0, float(t1.grad.abs().max()))
    for b in range(B):
        assert not E[b, lens[b, 0]:, :].any() and not E[b, :, lens[b, 1]:].any()
