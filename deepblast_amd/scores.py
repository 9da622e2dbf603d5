"""Score tensors for the DP on MI355X.

`alignment_scores(zx, zy, gx, gy)` replaces the two einsum + activation lines of the reference's
NeuralAligner (deepblast/alignment.py:122-123 and :134-135)

    theta = F.softplus(torch.einsum('bid,bjd->bij', zx, zy))
    A = F.logsigmoid(torch.einsum('bid,bjd->bij', gx, gy))

with one launch of a hand-written batched GEMM on the matrix cores whose epilogue applies the activation
(`sdp_scores_f32`, deepblast_amd/csrc/sdp_scores.hip: bf16 MFMA over exact three-piece operands, fp32 accuracy; the
f32-input MFMA for ragged D).  Differentiable: the backward needs no saved
pre-activations -- d softplus(s)/ds = sigmoid(s) = 1 - exp(-theta) and d logsigmoid(s)/ds = 1 - exp(A) -- and
forms the gradients of the embeddings with the same three-piece product (`sdp_scores_backward_f32`: the two contractions
per tensor run over the ROWS of the tensors as they lie in memory, no transposed copies); ragged widths (M or D not a
multiple of 4) are padded with zeros on the way in, unaligned views copied; only a few very large pairs -- too few tiles to
fill the chip -- are left to library GEMMs (torch.bmm).
"""
import torch

from . import _lib
from ._engine import get_engine, _ptr


class _Scores(torch.autograd.Function):

    @staticmethod
    def forward(ctx, zx, zy, gx, gy):
        eng = get_engine()
        dev = eng._dev(zx)
        eng._check(zx, zx=zx, zy=zy, gx=gx, gy=gy)
        if zx.dim() != 3 or zy.dim() != 3 or zx.shape[0] != zy.shape[0] or zx.shape[2] != zy.shape[2]:
            raise ValueError(f"zx must be (B,N,D) and zy (B,M,D); got {tuple(zx.shape)} and {tuple(zy.shape)}")
        if gx.shape != zx.shape or gy.shape != zy.shape:
            raise ValueError("gx / gy must have the shapes of zx / zy")
        zx_, zy_, gx_, gy_ = (t.detach().contiguous() for t in (zx, zy, gx, gy))
        B, N, D = zx_.shape
        M = zy_.shape[1]
        theta = torch.empty((B, N, M), dtype=torch.float32, device=zx.device)
        A = torch.empty((B, N, M), dtype=torch.float32, device=zx.device)
        # the library's own choice (sdp_api.hip): whole 16-deep slabs of 16-byte aligned rows take the three-piece bf16
        # product, anything else the f32-input MFMA kernel; the name only labels the launch for bench.py's timer
        x6 = D % 16 == 0 and all((t.data_ptr() & 15) == 0 for t in (zx_, zy_, gx_, gy_))
        t256 = -(-N // 256) * -(-M // 256)
        wide = (x6 and 16 * t256 <= 5 * (-(-N // 128) * -(-M // 128))
                and t256 * 2 * B >= 2 * torch.cuda.get_device_properties(dev).multi_processor_count)   # 256 x 256 tiles (sdp_api.hip)
        with torch.cuda.device(dev), eng._bracket(("sdp_scores_x6w_kernel" if wide else "sdp_scores_x6_kernel") if x6 else "sdp_scores_kernel"):
            rc = eng.lib.sdp_scores_f32(_ptr(zx_), _ptr(zy_), _ptr(gx_), _ptr(gy_), _ptr(theta), _ptr(A), B, N, M, D, dev,
                                        eng._stream(dev))
        _lib.check(rc, "sdp_scores_f32")
        ctx.save_for_backward(zx_, zy_, gx_, gy_, theta, A)
        return theta, A

    @staticmethod
    def backward(ctx, g_theta, g_A):
        zx, zy, gx, gy, theta, A = ctx.saved_tensors
        if _native_backward_ok(zx, zy, g_theta, g_A):
            return _native_backward(zx, zy, gx, gy, theta, A, g_theta, g_A)
        return _torch_backward(zx, zy, gx, gy, theta, A, g_theta, g_A)


def _torch_backward(zx, zy, gx, gy, theta, A, g_theta, g_A):
    """Library GEMMs (torch.bmm) -- what `_native_backward_ok` leaves to them: a few very large pairs, oversized grids."""
    out = [None, None, None, None]
    if g_theta is not None:
        ds = g_theta * (-torch.expm1(-theta))            # sigmoid(s) = 1 - exp(-softplus(s)); expm1: no cancellation for small theta
        out[0], out[1] = torch.bmm(ds, zy), torch.bmm(ds.transpose(1, 2), zx)
    if g_A is not None:
        ds = g_A * (-torch.expm1(A))                      # 1 - sigmoid(s) = 1 - exp(logsigmoid(s))
        out[2], out[3] = torch.bmm(ds, gy), torch.bmm(ds.transpose(1, 2), gx)
    return tuple(out)


def _native_backward_ok(zx, zy, g_theta, g_A):
    """Does the backward run on the native kernels?  Ragged widths (M or D not a multiple of 4) and unaligned views do -- padded /
    copied by `_native_backward`; what is left to the library GEMMs: a few very large pairs (below), grids beyond 65535, other
    dtypes or devices."""
    B, N, D = zx.shape
    M = zy.shape[1]
    M4, D4 = -(-M // 4) * 4, -(-D // 4) * 4
    if 2 * B > 65535 or max(N * M4, N * D4, M4 * D4) > (1 << 28):
        return False
    # one workgroup per 256 x 256 tile of an output: a few large pairs leave most CUs idle (4 x 2000 x 2000 x 256: 542 us
    # against 431 us for the library GEMMs; everything else measured was faster or equal, tools/scores_bwd_shapes.py)
    tiles = 2 * B * min(-(-N // 256), -(-M // 256)) * -(-D // 256)
    if tiles < torch.cuda.get_device_properties(zx.device).multi_processor_count // 2 and N * M > 512 * 512:
        return False
    return all(g is None or (g.dtype == torch.float32 and g.device == zx.device) for g in (g_theta, g_A))


def _pad_last(t, to):
    """(…, n) -> (…, to) with zeros behind (a fresh, aligned tensor); None stays None."""
    if t is None or t.shape[-1] == to:
        return t
    out = t.new_zeros(t.shape[:-1] + (to,))
    out[..., :t.shape[-1]] = t
    return out


def _pad_rows(t, to):
    """(B, n, D) -> (B, to, D) with zero rows behind."""
    if t is None or t.shape[1] == to:
        return t
    out = t.new_zeros((t.shape[0], to, t.shape[2]))
    out[:, :t.shape[1]] = t
    return out


def _native_backward(zx, zy, gx, gy, theta, A, g_theta, g_A):
    """sdp_scores_backward_f32: dS = g * dact/ds in one pass, then the two products per tensor on the bf16 pipe with exact
    three-piece operands (fp32 accuracy) -- one launch per side for both tensors (deepblast_amd/csrc/sdp_scores.hip).

    The kernels take M and D in multiples of 4 and 16-byte aligned tensors.  Ragged shapes -- the usual case: M is the longest
    sequence of the batch -- are padded with zeros here (columns of g, theta, A and rows of zy / gy for M; columns of the four
    embeddings for D: a zero cotangent column contributes nothing, a zero embedding column neither) and the results sliced;
    views that are not aligned are copied.  Round 5: until then such shapes went to torch.bmm."""
    eng = get_engine()
    dev = eng._dev(zx)
    B, N, D = zx.shape
    M = zy.shape[1]
    M4, D4 = -(-M // 4) * 4, -(-D // 4) * 4
    gt = None if g_theta is None else g_theta.contiguous()
    ga = None if g_A is None else g_A.contiguous()
    if M4 != M:
        gt, ga, theta, A = (_pad_last(t, M4) for t in (gt, ga, theta, A))
        zy, gy = _pad_rows(zy, M4), _pad_rows(gy, M4)
    if D4 != D:
        zx, zy, gx, gy = (_pad_last(t, D4) for t in (zx, zy, gx, gy))
    aligned = lambda t: t if (t is None or (t.is_contiguous() and (t.data_ptr() & 15) == 0)) else t.clone(memory_format=torch.contiguous_format)
    gt, ga, theta, A, zx, zy, gx, gy = (aligned(t) for t in (gt, ga, theta, A, zx, zy, gx, gy))
    ws = torch.empty(eng.lib.sdp_scores_backward_ws_bytes(B, N, M4) // 4, dtype=torch.float32, device=zx.device)
    new = lambda like: torch.empty_like(like)
    dzx, dzy = (new(zx), new(zy)) if gt is not None else (None, None)
    dgx, dgy = (new(gx), new(gy)) if ga is not None else (None, None)
    tensors = (gt, ga, theta, A, zx, zy, gx, gy, ws, dzx, dzy, dgx, dgy)
    if any(t is not None and (t.data_ptr() & 15) for t in tensors):   # (cannot happen with torch's allocator; never a silent wrong launch)
        raise RuntimeError("deepblast_amd.scores: a tensor of the native backward is not 16-byte aligned")
    with torch.cuda.device(dev), eng._bracket("sdp_scores_bwd"):
        rc = eng.lib.sdp_scores_backward_f32(*[_ptr(t) for t in tensors], B, N, M4, D4, dev, eng._stream(dev))
    _lib.check(rc, "sdp_scores_backward_f32")
    cut = lambda t, rows: None if t is None else (t[:, :rows, :D] if (t.shape[1] != rows or D4 != D) else t)
    return cut(dzx, N), cut(dzy, M), cut(dgx, N), cut(dgy, M)


def alignment_scores(zx, zy, gx, gy):
    """zx, gx: (B,N,D); zy, gy: (B,M,D) fp32 on one ROCm device -> (theta, A), each (B,N,M)."""
    return _Scores.apply(zx, zy, gx, gy)
