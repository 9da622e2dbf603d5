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
