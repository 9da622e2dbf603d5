"""Masked alignment losses on MI355X.

Drop-in for deepblast.losses.{MatrixCrossEntropy, SoftPathLoss, SoftAlignmentLoss}
(losses.py:9-48, 51-79, 82-118): same call signature `(first, Ypred, x_len, y_len, G)` and the same
scalar, differentiable w.r.t. `Ypred`.  The reference loops over the batch in Python (slice, masked_select,
reduce -- one host synchronisation per pair); here the forward is one launch (per-pair float64 sums) and
the backward one launch.
"""
import torch

from . import _lib
from ._engine import get_engine, _ptr

CROSS_ENTROPY, PATH, ALIGNMENT = 0, 1, 2


def _lens(x_len, y_len, B, device):
    x = torch.as_tensor(x_len, dtype=torch.int32).reshape(-1)
    y = torch.as_tensor(y_len, dtype=torch.int32).reshape(-1)
    if x.numel() != B or y.numel() != B:
        raise ValueError(f"x_len / y_len must have one entry per pair ({B})")
    return torch.stack([x, y], dim=1).to(device).contiguous()


class _MaskedLoss(torch.autograd.Function):

    @staticmethod
    def forward(ctx, first, pred, G, lens, kind):
        eng = get_engine()
        dev = eng._dev(pred)
        if pred.dtype != torch.float32:
            raise TypeError("HIP variant only supports torch.float32 type")
        first = first.detach().to(torch.float32).contiguous()
        G = G.detach().to(torch.float32).contiguous()
        p = pred.detach().contiguous()
        B, N, M = p.shape
        acc = torch.empty(B, dtype=torch.float64, device=p.device)
        cnt = torch.empty(B, dtype=torch.int32, device=p.device)
        with torch.cuda.device(dev), eng._bracket("sdp_loss_fwd_kernel"):
            rc = eng.lib.sdp_loss_forward_f32(_ptr(first), _ptr(p), _ptr(G), _ptr(lens), _ptr(acc), _ptr(cnt), B, N, M, kind,
                                              dev, eng._stream(dev))
        _lib.check(rc, "sdp_loss_forward_f32")
        if kind == CROSS_ENTROPY:
            per_pair = -(acc / cnt.to(torch.float64))          # -mean(pos + neg), losses.py:44
            scale = (-1.0 / (cnt.to(torch.float64) * B))
        else:
            per_pair = torch.sqrt(acc)                           # torch.norm of the masked vector
            sign = 1.0 if kind == PATH else -1.0
            scale = torch.where(per_pair > 0, sign / (per_pair * B), torch.zeros_like(per_pair))
        ctx.save_for_backward(first, p, G, lens, scale.to(torch.float32))
        ctx.kind = kind
        return (per_pair.sum() / B).to(torch.float32)

    @staticmethod
    def backward(ctx, gout):
        first, p, G, lens, scale = ctx.saved_tensors
        eng = get_engine()
        dev = eng._dev(p)
        B, N, M = p.shape
        grad = torch.empty_like(p)
        sc = (scale * gout.to(torch.float32)).contiguous()
        with torch.cuda.device(dev), eng._bracket("sdp_loss_bwd_kernel"):
            rc = eng.lib.sdp_loss_backward_f32(_ptr(first), _ptr(p), _ptr(G), _ptr(lens), _ptr(sc), _ptr(grad), B, N, M,
                                               ctx.kind, dev, eng._stream(dev))
        _lib.check(rc, "sdp_loss_backward_f32")
        return None, grad, None, None, None


class _Loss:
    kind = None

    def __call__(self, first, Ypred, x_len, y_len, G):
        lens = _lens(x_len, y_len, Ypred.shape[0], Ypred.device)
        return _MaskedLoss.apply(first, Ypred, G, lens, self.kind)


class MatrixCrossEntropy(_Loss):
    """-(mean_G(Ytrue log p) + mean_G((1-Ytrue) log(1-p))) per pair, averaged over pairs (losses.py:9-48)."""
    kind = CROSS_ENTROPY


class SoftPathLoss(_Loss):
    """|| (P * Ypred)[G] ||_2 per pair, averaged over pairs (losses.py:51-79); first argument is P."""
    kind = PATH


class SoftAlignmentLoss(_Loss):
    """|| (Ytrue - Ypred)[G] ||_2 per pair, averaged over pairs (losses.py:82-118)."""
    kind = ALIGNMENT


# ----------------------------------------------------------------------------------------------------------------
# decode + loss as ONE differentiable op ("masked loss on E fused with the adjoint seed")
# ----------------------------------------------------------------------------------------------------------------
class _DecodeLoss(torch.autograd.Function):
    """loss(decode(theta, A)) with the loss's gradient fed to the adjoint forward sweep inside the kernel.

    The unfused training step (reference: alignment.py:124 then losses.py, trainer.py:154-171) runs
    forward -> backward (E) -> loss -> loss backward (writes dLoss/dE, a (B,N,M) tensor) -> adjoint forward (reads it)
    -> adjoint backward.  Here the loss's backward kernel and its tensor disappear: `sdp_adjoint_forward_loss_f32`
    forms dLoss/dE from (first, E, G, scale) while it stages them.  Same values as the unfused path."""

    @staticmethod
    def forward(ctx, theta, A, first, G, lens_loss, lens_dp, kind, variant, fill=False):
        from ._dp import _validate
        _validate(theta, A, 'softmax', False)
        if theta.dtype != torch.float32:
            # the loss kernels read E through a raw float32 pointer: a float64 E (the engine's f64 path) must never get there
            raise TypeError(f"decode_loss supports torch.float32 tensors only, got {theta.dtype}; with float64 use "
                            "loss(first, decoder.decode(theta, A), x_len, y_len, G)")
        eng = get_engine()
        dev = eng._dev(theta)
        first = first.detach().to(torch.float32).contiguous()
        G = G.detach().to(torch.float32).contiguous()
        th, a = theta.detach(), A.detach()
        B, N, M = th.shape
        Vt, Q = eng.forward(th, a, variant, lens_dp, exact_state=True)
        ones = torch.ones(B, dtype=torch.float32, device=th.device)
        # E outside the pairs' blocks is read by nobody in this op WHEN THE LOSS MASKS BY THE DP'S OWN LENGTHS (one tensor for
        # both: the loss slices by it, both adjoint sweeps mask by it): then it is not filled unless the caller wants the returned
        # E whole.  If the loss's lengths are another tensor (x_len / y_len may exceed `lengths`: the API allows it) the loss kernel
        # and the per-pair scale would read E between the two blocks, so E is zero-filled as decode() fills it.
        lean = (not fill) and lens_dp is not None and lens_dp is lens_loss
        E = eng.backward(ones, Q, (B, N, M), variant, lens_dp, exact_state=True, **({"no_fill": True} if lean else {}))
        acc = torch.empty(B, dtype=torch.float64, device=th.device)
        cnt = torch.empty(B, dtype=torch.int32, device=th.device)
        with torch.cuda.device(dev), eng._bracket("sdp_loss_fwd_kernel"):
            rc = eng.lib.sdp_loss_forward_f32(_ptr(first), _ptr(E), _ptr(G), _ptr(lens_loss), _ptr(acc), _ptr(cnt), B, N, M, kind,
                                              dev, eng._stream(dev))
        _lib.check(rc, "sdp_loss_forward_f32")
        if kind == CROSS_ENTROPY:
            per_pair = -(acc / cnt.to(torch.float64))
            scale = (-1.0 / (cnt.to(torch.float64) * B))
        else:
            per_pair = torch.sqrt(acc)
            sign = 1.0 if kind == PATH else -1.0
            scale = torch.where(per_pair > 0, sign / (per_pair * B), torch.zeros_like(per_pair))
        ctx.save_for_backward(Q, E, first, G, lens_loss, scale.to(torch.float32))
        ctx.others = (kind, variant, lens_dp, lens_dp is lens_loss)   # (saved tensors come back as new objects: remember the identity)
        ctx.mark_non_differentiable(E)
        return (per_pair.sum() / B).to(torch.float32), E

    @staticmethod
    def backward(ctx, gout, _gE):
        Q, E, first, G, lens_loss, scale = ctx.saved_tensors
        kind, variant, lens_dp, same_lens = ctx.others
        eng = get_engine()
        sc = (scale * gout.to(torch.float32)).contiguous()
        # the loss masks with its own lengths; the DP sweeps use theirs (None = full padded matrix, as the reference)
        if lens_loss is not None and not same_lens:
            # cells outside the LOSS's block must not seed the sweep (the kernel masks with the DP's lengths only): fold
            # the loss's lengths into the mask -- unless both are the same tensor (decode_loss passes one object when
            # `lengths` equals (x_len, y_len), the usual case)
            B, N, M = E.shape
            ii = torch.arange(N, device=E.device).view(1, N, 1) < lens_loss[:, 0].view(B, 1, 1)
            jj = torch.arange(M, device=E.device).view(1, 1, M) < lens_loss[:, 1].view(B, 1, 1)
            G = G * (ii & jj).to(G.dtype)
        _, Qd = eng.adjoint_forward_loss(Q, first, E, G, sc, kind, variant, lens_dp)
        Ed = eng.adjoint_backward(E, Q, Qd, variant, lens_dp)
        return Ed, None, None, None, None, None, None, None, None


def decode_loss(decoder, loss, theta, A, first, x_len, y_len, G, lengths=None, fill=False):
    """`loss(first, decoder.decode(theta, A[, lengths]), x_len, y_len, G)` as one op -> (loss scalar, E).

    decoder : NeedlemanWunschDecoder / SmithWatermanDecoder of this package
    loss    : MatrixCrossEntropy() / SoftPathLoss() / SoftAlignmentLoss() of this module (its `kind` is used)
    lengths : optional (B,2) per-pair sizes for the DP itself (None = the reference's full padded DP)
    The scalar is differentiable w.r.t. theta (the gradient w.r.t. A is None, as in the reference's second-order
    path, nw.py:386); E is returned for inspection / traceback and is not differentiable through this op.
    fill    : with `lengths` EQUAL to (x_len, y_len) -- the usual case -- False (default) leaves E OUTSIDE each pair's block
              unwritten (uninitialised memory, possibly NaN): nothing in this op reads it (the loss slices by x_len / y_len as
              deepblast/losses.py:30-40 does, the sweeps mask by `lengths`), and `decoder.traceback_batch(E, lengths)` does not
              either; True zero-fills it like decoder.decode() does.  When `lengths` differs from (x_len, y_len) E is always
              zero-filled: the loss then reads cells outside the DP's blocks.  The gradient w.r.t. theta is always zero
              outside the blocks."""
    from ._engine import NW, SW
    from .sw import SmithWatermanDecoder
    variant = SW if isinstance(decoder, SmithWatermanDecoder) else NW
    if getattr(decoder, "arithmetic", "fast") != "fast":
        raise NotImplementedError("decode_loss runs the tuned sweeps only; with arithmetic='reference' use "
                                  "loss(first, decoder.decode(theta, A), x_len, y_len, G)")
    B = theta.shape[0]
    lens_loss = _lens(x_len, y_len, B, theta.device)
    lens_dp = None
    if lengths is not None:
        same = False
        if not isinstance(lengths, torch.Tensor) and not isinstance(x_len, torch.Tensor) and not isinstance(y_len, torch.Tensor):
            import numpy as _np
            la = _np.asarray(lengths)
            same = la.shape == (B, 2) and _np.array_equal(la[:, 0], _np.asarray(x_len)) and _np.array_equal(la[:, 1], _np.asarray(y_len))
        lens_dp = lens_loss if (same and lens_loss is not None) else get_engine()._lens(lengths, B, theta.device)
    return _DecodeLoss.apply(theta, A, first, G, lens_loss, lens_dp, loss.kind, variant, bool(fill))
