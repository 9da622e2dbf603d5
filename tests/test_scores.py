"""Scores: CPU -- the numpy oracle against the fixture written from the real torch ops
(alignment.py:122-123); GPU -- the MFMA kernel (sdp_scores_f32) against fixture and oracle, gradients included.
Tolerance 1e-4 relative to max(1, |ref|) (north_star); observed ~1e-6."""
import os

import numpy as np
import pytest

import datagen
import parity
from oracle import scores_oracle


def _cases(golden_dir):
    d = np.load(os.path.join(golden_dir, "g11_scores.npz"))
    for idx in range(len(d["shapes"])):
        yield idx, {k[len(f"s{idx}_"):]: d[k] for k in d.files if k.startswith(f"s{idx}_")}


def test_oracle_matches_the_reference_ops(golden_dir):
    for idx, c in _cases(golden_dir):
        theta, A = scores_oracle.scores(c["zx"], c["zy"], c["gx"], c["gy"])
        # the fixture is torch's fp32 einsum (its own summation order): agreement to fp32 round-off of the inner product
        assert parity.rel_err(theta, c["theta"]) <= 2e-6, idx
        assert parity.rel_err(A, c["A"]) <= 2e-6, idx
    s = np.array([-100.0, -20.0, -1e-3, 0.0, 1e-3, 19.9, 20.1, 100.0])
    assert np.allclose(scores_oracle.softplus(s), np.logaddexp(0, s), rtol=1e-9, atol=0)
    assert np.allclose(scores_oracle.logsigmoid(s), -np.logaddexp(0, -s), rtol=1e-9, atol=0)


@pytest.mark.gpu
def test_kernel_matches_fixture_and_gradients(golden_dir):
    import torch
    from deepblast_amd.scores import alignment_scores
    for idx, c in _cases(golden_dir):
        t = [torch.from_numpy(c[k]).cuda().requires_grad_() for k in ("zx", "zy", "gx", "gy")]
        theta, A = alignment_scores(*t)
        ((theta * torch.from_numpy(c["wt"]).cuda()).sum() + (A * torch.from_numpy(c["wa"]).cuda()).sum()).backward()
        assert parity.rel_err(theta.detach().cpu().numpy(), c["theta"]) <= parity.TOL, idx
        assert parity.rel_err(A.detach().cpu().numpy(), c["A"]) <= parity.TOL, idx
        for k, a in zip(("dzx", "dzy", "dgx", "dgy"), t):
            assert parity.abs_err(a.grad.cpu().numpy(), c[k], scale=True) <= parity.TOL, (idx, k)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(256, 512, 512, 512), (3, 127, 129, 31), (2, 300, 5, 100), (1, 1, 2048, 7),
                                   # D a multiple of 16: the three-piece bf16 kernel, ragged tiles in N and M
                                   (3, 127, 129, 48), (2, 300, 5, 64), (1, 1, 2048, 16), (2, 130, 257, 1024)],
                         ids=lambda s: "x".join(map(str, s)))
def test_kernel_matches_oracle(shape):
    import torch
    from deepblast_amd.scores import alignment_scores
    B, N, M, D = shape
    Bo = min(B, 4)   # the float64 oracle on a few pairs of a full batch (the rest: batch independence, below)
    sc = 2.0 / np.sqrt(D)
    arrs = [datagen.normal(600 + i, (B, n, D)) * (sc if i % 2 == 0 else 2 * sc) for i, n in enumerate((N, M, N, M))]
    t = [torch.from_numpy(a.astype(np.float32)).cuda() for a in arrs]
    theta, A = alignment_scores(*t)
    sel = np.linspace(0, B - 1, Bo).astype(int)
    rt, ra = scores_oracle.scores(*[a.astype(np.float32)[sel] for a in arrs])
    assert parity.rel_err(theta[sel].cpu().numpy(), rt) <= parity.TOL
    assert parity.rel_err(A[sel].cpu().numpy(), ra) <= parity.TOL
    # a pair's scores do not depend on the batch it is computed in (bit-exact)
    alone_t, alone_a = alignment_scores(*[x[sel[-1]:sel[-1] + 1].contiguous() for x in t])
    assert torch.equal(alone_t[0], theta[sel[-1]]) and torch.equal(alone_a[0], A[sel[-1]])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(160, 500, 760, 48), (150, 510, 514, 32), (256, 512, 512, 512)], ids=lambda s: "x".join(map(str, s)))
def test_wide_and_narrow_tiles_agree_bit_for_bit(shape):
    """Full batches take the 256 x 256-tile kernel (sdp_scores_x6w_kernel: 8 waves, epilogue through LDS), small ones the
    128 x 128 tiles; same arithmetic in the same order, so the results must be identical -- ragged edges in N and M, an M
    that is not a multiple of 4 (the epilogue's dword path) included.  The experiments build can force the small tiles."""
    import ctypes
    import os
    import torch
    from deepblast_amd import _lib, build
    B, N, M, D = shape
    t256 = -(-N // 256) * -(-M // 256)
    assert 16 * t256 <= 5 * (-(-N // 128) * -(-M // 128)) and t256 * 2 * B >= 512   # this shape does take the wide kernel on 256 CUs
    exp = _lib.load_path(build.EXP_OUT)
    t = [torch.from_numpy((datagen.normal(800 + i, (B, n, D)) * 2.0 / np.sqrt(D)).astype(np.float32)).cuda() for i, n in enumerate((N, M, N, M))]
    outs = []
    for mask in (0, 32):
        exp.sdp_set_debug(mask)
        theta = torch.full((B, N, M), 7.0, device="cuda")
        A = torch.full((B, N, M), 7.0, device="cuda")
        rc = exp.sdp_scores_f32(*(x.data_ptr() for x in t), theta.data_ptr(), A.data_ptr(), B, N, M, D, 0, torch.cuda.current_stream().cuda_stream)
        exp.sdp_set_debug(0)
        assert rc == 0
        outs.append((theta, A))
    torch.cuda.synchronize()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    sel = [0, B // 2, B - 1]
    rt, ra = scores_oracle.scores(*[x[sel].cpu().numpy() for x in t])
    assert parity.rel_err(outs[0][0][sel].cpu().numpy(), rt) <= parity.TOL and parity.rel_err(outs[0][1][sel].cpu().numpy(), ra) <= parity.TOL


@pytest.mark.gpu
def test_bf16_piece_kernel_has_fp32_accuracy_and_unaligned_rows_fall_back():
    """The default kernel for D % 16 == 0 multiplies exact three-piece bf16 operands (six products per k): its error
    against a float64 einsum must be that of an fp32 product (a two-piece split would show 2^-16 per product: 3e-4 here),
    also on large scores; embeddings that are not 16-byte aligned take the f32-input kernel and agree to the same bound."""
    import torch
    import torch.nn.functional as F
    from deepblast_amd.scores import alignment_scores
    B, N, M, D = 6, 200, 333, 512
    for scale, bound in ((1.0, 1e-6), (8.0, 3e-5)):
        t = [torch.from_numpy((datagen.normal(700 + i, (B, n, D)) * scale / np.sqrt(D)).astype(np.float32)).cuda() for i, n in enumerate((N, M, N, M))]
        ref_t = F.softplus(torch.einsum("bid,bjd->bij", t[0].double(), t[1].double()))
        ref_a = F.logsigmoid(torch.einsum("bid,bjd->bij", t[2].double(), t[3].double()))
        theta, A = alignment_scores(*t)
        assert float((theta.double() - ref_t).abs().max()) <= bound and float((A.double() - ref_a).abs().max()) <= bound
        # the same embeddings at an address that is 4 (not 16) bytes aligned
        shifted = []
        for x in t:
            buf = torch.empty(x.numel() + 1, dtype=torch.float32, device="cuda")
            buf[1:].copy_(x.reshape(-1))
            shifted.append(buf[1:].view_as(x))
            assert shifted[-1].data_ptr() % 16 == 4 and shifted[-1].is_contiguous()
        theta2, A2 = alignment_scores(*shifted)
        assert float((theta2.double() - ref_t).abs().max()) <= bound and float((A2.double() - ref_a).abs().max()) <= bound


@pytest.mark.gpu
def test_scores_feed_the_dp():
    """alignment.py:122-124 end to end: embeddings -> theta, A (MFMA kernel) -> decode (DP sweeps), against the oracles."""
    import torch
    from deepblast_amd import NeedlemanWunschDecoder
    from deepblast_amd.scores import alignment_scores
    B, N, M, D = 3, 70, 90, 64
    arrs = [(datagen.normal(700 + i, (B, n, D)) / np.sqrt(D) * 2).astype(np.float32) for i, n in enumerate((N, M, N, M))]
    theta, A = alignment_scores(*[torch.from_numpy(a).cuda() for a in arrs])
    theta, A = theta.detach().requires_grad_(), A.detach().requires_grad_()   # decode differentiates w.r.t. both
    aln = NeedlemanWunschDecoder("softmax").decode(theta, A)
    rt, ra = scores_oracle.scores(*arrs)
    ref = parity.oracle_all(rt, ra, None, None, 0)
    assert parity.abs_err(aln.detach().cpu().numpy(), ref["E"]) <= parity.TOL


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(256, 512, 512, 512), (3, 130, 260, 48), (2, 300, 128, 100), (1, 1, 2048, 8), (5, 257, 516, 20), (4, 600, 36, 64),
                                   (2, 50, 33, 24), (3, 70, 129, 30), (2, 65, 255, 7), (1, 3, 1, 1)],   # (round 5: ragged M / D, padded on the way in)
                         ids=lambda s: "x".join(map(str, s)))
def test_native_backward_matches_float64_autograd(shape):
    """sdp_scores_backward_f32 (dS pass + two three-piece products per tensor whose contractions run over the rows of the
    tensors in memory) against float64 autograd through the reference's own ops (alignment.py:122-123) -- fp32 accuracy --
    and against the library-GEMM path it replaces; one-sided gradients (theta only / A only) go through the same kernels."""
    import torch
    import torch.nn.functional as F
    from deepblast_amd import scores as sc
    B, N, M, D = shape
    t = [torch.from_numpy((datagen.normal(800 + i, (B, n, D)) * 2.0 / np.sqrt(D)).astype(np.float32)).cuda() for i, n in enumerate((N, M, N, M))]
    wt = torch.from_numpy(datagen.normal(810, (B, N, M)).astype(np.float32)).cuda()
    wa = torch.from_numpy(datagen.normal(811, (B, N, M)).astype(np.float32)).cuda()
    assert sc._native_backward_ok(t[0], t[1], wt, wa)
    t64 = [x.double().requires_grad_() for x in t]
    th64 = F.softplus(torch.einsum("bid,bjd->bij", t64[0], t64[1]))
    a64 = F.logsigmoid(torch.einsum("bid,bjd->bij", t64[2], t64[3]))
    ((th64 * wt.double()).sum() + (a64 * wa.double()).sum()).backward()
    ref = [x.grad for x in t64]
    calls = []
    orig = sc._native_backward
    sc._native_backward = lambda *a: (calls.append(1), orig(*a))[1]
    try:
        for mode in ("both", "theta", "A"):
            tt = [x.clone().requires_grad_() for x in t]
            theta, A = sc.alignment_scores(*tt)
            loss = (theta * wt).sum() * (mode != "A") + (A * wa).sum() * (mode != "theta")
            if mode == "both":
                loss.backward()
                got = [x.grad for x in tt]
            elif mode == "theta":
                got = list(torch.autograd.grad((theta * wt).sum(), tt[:2])) + [None, None]
            else:
                got = [None, None] + list(torch.autograd.grad((A * wa).sum(), tt[2:]))
            for k, (g, r) in enumerate(zip(got, ref)):
                if g is None:
                    continue
                scale = max(1.0, float(r.abs().max()))
                err = float((g.double() - r).abs().max()) / scale
                assert err <= 2e-6, (shape, mode, k, err)   # an fp32 sum of <= 2048 products of this size: ~1e-7 .. 1e-6
    finally:
        sc._native_backward = orig
    assert len(calls) == 3
    # the path it replaces (torch.bmm) agrees to the same bound
    th, A = sc.alignment_scores(*t)
    lib = sc._torch_backward(t[0], t[1], t[2], t[3], th, A, wt, wa)
    nat = sc._native_backward(t[0], t[1], t[2], t[3], th, A, wt, wa)
    for k in range(4):
        scale = max(1.0, float(lib[k].abs().max()))
        assert float((lib[k] - nat[k]).abs().max()) / scale <= 5e-6, (shape, k)


@pytest.mark.gpu
def test_backward_ragged_shapes_and_views_run_native_and_what_is_left_to_the_library():
    """Round 5: M or D not a multiple of 4 and unaligned views no longer leave the native kernels (padded / copied in
    `_native_backward`; the C entry point itself still refuses them); the library GEMMs keep a few very large pairs."""
    import torch
    import torch.nn.functional as F
    from deepblast_amd import scores as sc
    B, N, M, D = 2, 50, 33, 24   # M not a multiple of 4
    base = [torch.from_numpy((datagen.normal(820 + i, (B, n + 1, D + 1)) / np.sqrt(D)).astype(np.float32)).cuda() for i, n in enumerate((N, M, N, M))]
    views = [b[:, 1:, 1:] for b in base]               # neither contiguous nor 16-byte aligned
    assert all((v.data_ptr() & 15) != 0 and not v.is_contiguous() for v in views)
    t = [v.detach().requires_grad_() for v in views]
    assert sc._native_backward_ok(t[0], t[1], None, None)
    calls = []
    orig = sc._native_backward
    sc._native_backward = lambda *a: (calls.append(1), orig(*a))[1]
    try:
        theta, A = sc.alignment_scores(*t)
        (theta.sum() + A.sum()).backward()
    finally:
        sc._native_backward = orig
    assert calls == [1]
    t64 = [v.detach().double().requires_grad_() for v in views]
    (F.softplus(torch.einsum("bid,bjd->bij", t64[0], t64[1])).sum() + F.logsigmoid(torch.einsum("bid,bjd->bij", t64[2], t64[3])).sum()).backward()
    for x, r in zip(t, t64):
        assert x.grad is not None and x.grad.shape == r.grad.shape
        assert float((x.grad.double() - r.grad).abs().max()) <= 2e-6 * max(1.0, float(r.grad.abs().max()))
    # the C entry point takes multiples of 4 only: an error code, not a launch
    eng_lib = sc.get_engine().lib
    ws = torch.empty(8, device="cuda")
    tc = [x.detach().contiguous() for x in t]
    rc = eng_lib.sdp_scores_backward_f32(theta.data_ptr(), None, theta.data_ptr(), None, tc[0].data_ptr(), tc[1].data_ptr(), None, None, ws.data_ptr(),
                                         tc[0].data_ptr(), tc[1].data_ptr(), None, None, B, N, M, D, 0, None)
    assert rc != 0   # SDP_E_SHAPE
    # a few very large pairs: too few tiles for the chip, left to the library GEMMs
    big = [torch.empty(4, n, 256, device="cuda") for n in (2000, 2000)]
    assert not sc._native_backward_ok(big[0], big[1], None, None)
