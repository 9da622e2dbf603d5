"""Autograd wiring shared by the Needleman-Wunsch and Smith-Waterman operators.

Mirrors the reference's Function pair (deepblast/nw_cuda.py:168-262, nw.py:315-386):

    Function.forward(theta, A, operator)            -> Vt            saves (theta, A, state)
    Function.backward(Et)                           -> (E, A, None)  via FunctionBackward.apply
    FunctionBackward.forward(theta, A, Et, Q, op)   -> (E, A)        saves (Q, E) or (theta, A, E)
    FunctionBackward.backward(Ztheta, ZA)           -> (Ed, None, Vtd, None, None)

and keeps its gradient-flow quirks: the first-order "gradient" returned
for A is A itself (nw.py:337-339,355); the second-order gradient w.r.t. A is None
(nw.py:386); Et may be non-uniform.

Differences that are part of the design, not of the maths: `Q` is an opaque state tensor
(library-private layout; 5 bytes per cell -- two 20-bit weights -- on the inference path, float2 when decode() announces
that the second-order sweeps will follow) instead of (B,N+2,M+2,3), and E is produced directly
as (B,N,M) -- the reference's E[:,1:-1,1:-1] -- without materialising the zero border.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _engine


def _validate(theta, A, operator, allow_none_operator):
    # error behaviour of the reference GPU variant (nw_cuda.py:171-175)
    if operator != 'softmax' and not (allow_none_operator and operator is None):
        raise NotImplementedError("HIP variant only supports 'softmax' operator")
    # float32 (the reference's GPU classes take nothing else, nw_cuda.py:174-175) or float64 (its CPU classes take what they
    # are given, and its own tests hand them float64: tests/test_nw.py:46-90) -- both tensors alike
    if theta.dtype not in (torch.float32, torch.float64) or A.dtype != theta.dtype:
        raise TypeError(f"HIP variant supports torch.float32 (and, unoptimised, torch.float64) tensors of one dtype; got {theta.dtype} and {A.dtype}")
    if theta.dim() != 3 or A.shape != theta.shape:
        raise ValueError(f"theta and A must both be (B, N, M); got {tuple(theta.shape)} and {tuple(A.shape)}")
    if A.device != theta.device:
        # the kernels receive raw pointers: a tensor on another device would be a foreign address
        raise ValueError(f"theta and A must live on the same device; got {theta.device} and {A.device}")


def _same_device(ref, **others):
    """Raw pointers cross the C ABI: every tensor of a call must be fp32 on the device the kernel runs on."""
    for name, t in others.items():
        if t is None:
            continue
        if t.device != ref.device:
            raise ValueError(f"{name} is on {t.device}, expected {ref.device}")
        if t.dtype != ref.dtype:
            raise TypeError(f"{name} must be {ref.dtype}, got {t.dtype}")


def make_functions(variant, prefix, allow_none_operator=False):
    """Build the (Function, FunctionBackward) pair for one variant."""

    class FunctionBackward(torch.autograd.Function):

        @staticmethod
        def forward(ctx, theta, A, Et, Q, operator, lens=None, exact_state=False, no_fill=False):
            eng = _engine.get_engine()
            if Et.device != theta.device:
                raise ValueError(f"Et is on {Et.device}, expected {theta.device}")
            E = eng.backward(Et.detach(), Q, tuple(theta.shape), variant, lens, exact_state=exact_state, **({"no_fill": True} if no_fill else {}))
            # exact state: the adjoint sweeps can use Q as it is; compact state: they need theta and A to get it
            if exact_state:
                ctx.save_for_backward(Q, E)
            else:
                ctx.save_for_backward(theta, A, E)
            ctx.others = (operator, lens, exact_state)
            # The cotangent of the pass-through A output is all zeros whenever nothing consumes it (the
            # reference materialises it, nw.py:357-383, and feeds the zeros to the adjoint sweep).  Asking
            # autograd not to materialise lets the kernel skip reading a (B,N,M) tensor of zeros.
            ctx.set_materialize_grads(False)
            return E, A

        @staticmethod
        def backward(ctx, Ztheta, ZA):
            _, lens, exact_state = ctx.others
            eng = _engine.get_engine()
            if exact_state:
                Q, E = ctx.saved_tensors
            else:
                # The saved state is the compact one (5 B/cell) the backward sweep reads fastest.  The adjoint
                # sweeps multiply the weights with directional derivatives of any magnitude and need them at
                # full fp32 precision: re-run the forward sweep in its exact-state form.  Callers that know
                # the second-order sweeps will follow (Decoder.decode, i.e. training) ask for the exact state
                # up front and never get here.
                theta, A, E = ctx.saved_tensors
                _, Q = eng.forward(theta.detach(), A.detach(), variant, lens, exact_state=True)
            if Ztheta is None:
                Ztheta = torch.zeros_like(E)
            _same_device(E, Ztheta=Ztheta, ZA=ZA)
            ref = exact_state == _engine.REF
            Vtd, Qd = eng.adjoint_forward(Q, Ztheta, ZA, variant, lens, ref=ref)
            Ed = eng.adjoint_backward(E, Q, Qd, variant, lens, ref=ref)
            return Ed, None, Vtd, None, None, None, None, None

    class Function(torch.autograd.Function):

        @staticmethod
        def forward(ctx, theta, A, operator, lens=None, exact_state=False, no_fill=False):
            _validate(theta, A, operator, allow_none_operator)
            if theta.dtype == torch.float64:
                exact_state = _engine.F64   # (truthy: the state serves all four sweeps, as with exact_state=True)
            eng = _engine.get_engine()
            Vt, Q = eng.forward(theta.detach(), A.detach(), variant, lens, exact_state=exact_state)
            ctx.save_for_backward(theta, A, Q)
            ctx.others = (operator, lens, exact_state, no_fill)
            return Vt

        @staticmethod
        def backward(ctx, Et):
            theta, A, Q = ctx.saved_tensors
            operator, lens, exact_state, no_fill = ctx.others
            E, A = FunctionBackward.apply(theta, A, Et, Q, operator, lens, exact_state, no_fill)
            return E, A, None, None, None, None

    Function.__name__ = Function.__qualname__ = prefix + "Function"
    FunctionBackward.__name__ = FunctionBackward.__qualname__ = prefix + "FunctionBackward"
    return Function, FunctionBackward


def traceback(grad, rule="cpu"):
    """Greedy arg-max walk over one (N, M) expected-alignment matrix -> [(i, j, state)].

    rule="cpu" (default, the parity oracle's class) is described below; rule="cuda" is the walk of the reference's
    GPU classes (deepblast/nw_cuda.py:273-317, sw_cuda.py:283-327), the classes this package replaces: it stops as soon
    as ANY of the three neighbours is off the matrix (`or` instead of `and`, nw_cuda.py:297) or equals its sentinel
    -1e10, so it never wraps and never raises; the two differ when a walk reaches row 0 or column 0 early.

    Same rule as the reference's CPU decoder (deepblast/nw.py:401-444, sw.py:328-371):
    start at the bottom-right match, repeatedly step to the largest of
    left=(i-1,j) [state x=0], diag=(i-1,j-1) [m=1], upper=(i,j-1) [y=2] (first wins ties),
    stop when all three are off the matrix, then pad the remaining gaps.  The reference
    reads grad[i-1, j-1] with Python's negative-index wrap when exactly one of i, j is 0
    and can walk off the matrix (IndexError) on inputs that are not alignment matrices;
    both behaviours are preserved.
    """
    if rule not in ("cpu", "cuda"):
        raise ValueError(f"traceback rule must be 'cpu' or 'cuda', got {rule!r}")
    x, m, y = 0, 1, 2
    g = grad.detach().cpu().numpy() if isinstance(grad, torch.Tensor) else np.asarray(grad)
    N, M = g.shape
    floor = -1e10 if rule == "cuda" else -100000
    i, j = N - 1, M - 1
    states = [(i, j, m)]
    while True:
        left = floor if i <= 0 else g[i - 1, j]
        diag = floor if (i <= 0 and j <= 0) else g[i - 1, j - 1]
        upper = floor if j <= 0 else g[i, j - 1]
        if rule == "cuda":
            if left == floor or diag == floor or upper == floor:
                break
        elif left == floor and diag == floor and upper == floor:
            break
        # the reference compares through torch.Tensor([...]), i.e. in float32, first maximum wins
        cands = (np.float32(left), np.float32(diag), np.float32(upper))
        best = 0
        for k in (1, 2):
            if cands[k] > cands[best]:
                best = k
        i, j = ((i - 1, j), (i - 1, j - 1), (i, j - 1))[best]
        states.append((i, j, (x, m, y)[best]))
    while i > 0:
        i -= 1
        states.append((i, j, x))
    while j > 0:
        j -= 1
        states.append((i, j, y))
    return states[::-1]


class _Decoder(nn.Module):
    """Common body of NeedlemanWunschDecoder / SmithWatermanDecoder (nw_cuda.py:265-325)."""

    _function = None

    def __init__(self, operator, traceback_rule="cpu", arithmetic="fast"):
        """traceback_rule (extension): "cpu" = the walk of the reference's CPU decoders (nw.py:401-444, the parity
        oracle), "cuda" = the walk of its GPU decoders (nw_cuda.py:273-317), for callers that switch over from those.
        arithmetic (extension): "fast" = the tuned sweeps (fp32 exp-domain forward, float64 products in the second-order
        pair: within 1e-4 of the reference wherever the reference is within 1e-4 of its own float64 run, and closer to
        that float64 run than the reference is); "reference" = the reference's arithmetic rounding for rounding
        (include/sdp.h: SDP_REF_ROUNDING) -- unoptimised, for callers who need nw.py's numbers on long saturated
        alignments, where nw.py's own fp32 roundings move the second-order results by 1-2e-4."""
        super().__init__()
        if traceback_rule not in ("cpu", "cuda"):
            raise ValueError(f"traceback_rule must be 'cpu' or 'cuda', got {traceback_rule!r}")
        if arithmetic not in ("fast", "reference"):
            raise ValueError(f"arithmetic must be 'fast' or 'reference', got {arithmetic!r}")
        self.operator = operator
        self.traceback_rule = traceback_rule
        self.arithmetic = arithmetic

    def forward(self, theta, A, lengths=None, fill=True):
        """theta, A: (B, N, M) fp32 on a ROCm device -> Vt (B,) on the same device.

        `lengths` (optional, (B,2) int) is an extension: per-pair true sizes of a padded
        batch; None reproduces the reference (DP over the full padded matrix).
        `fill` (with lengths): True = the gradient E is zero outside each pair's n_b x m_b block (the contract);
        False = those cells are NOT written and hold whatever the allocator handed out (include/sdp.h: SDP_NO_FILL) --
        for callers that mask by the same lengths (a loss that slices [:x_len, :y_len], `traceback_batch(E, lengths)`):
        the zero fill of a padded batch moves as many bytes as the sweep itself."""
        tr = self._transposed(theta, A, lengths)
        if tr is not None:
            theta, A, lengths = tr
        if self.arithmetic == "reference":
            return self._function.apply(theta, A, self.operator, lengths, _engine.REF)
        if lengths is None:
            return self._function.apply(theta, A, self.operator)
        if not fill:
            return self._function.apply(theta, A, self.operator, lengths, False, True)
        return self._function.apply(theta, A, self.operator, lengths)

    @staticmethod
    def _transposed(theta, A, lengths):
        """More columns than the sweeps take (the boundary rows of a strip live in LDS: sdp_max_cols() = 2048, the limit of the
        reference's GPU classes, nw_cuda.py:11) but not more rows: the recurrence is symmetric in its two axes -- V[i,j] =
        theta[i,j] + lse(A[i,j] + V[i-1,j], V[i-1,j-1], A[i,j] + V[i,j-1]), one gap score for both directions (nw.py:46-62) --
        so the problem is swept on the TRANSPOSED tensors (N has no limit) and autograd transposes every gradient back: E, the
        pass-through A, the second-order results.  The parity oracle nw.py has no column limit.  -> (theta^T, A^T, lengths
        with their columns swapped), or None when no transposition is needed (or would not help: both sides too long)."""
        if theta.dim() != 3 or A.shape != theta.shape:
            return None
        cap = _engine.get_engine().max_cols()
        if theta.shape[2] <= cap or theta.shape[1] > cap:
            return None
        if lengths is not None:
            lengths = torch.as_tensor(lengths)
            lengths = torch.stack([lengths[:, 1], lengths[:, 0]], dim=1)
        return theta.transpose(1, 2), A.transpose(1, 2), lengths

    def _forward_for_decode(self, theta, A, lengths, fill=True):
        tr = self._transposed(theta, A, lengths)
        if tr is not None:
            theta, A, lengths = tr
        # decode() is differentiated again by its callers (training: loss on the alignment matrix): save the
        # state in the exact form all four sweeps can share
        xs = _engine.REF if self.arithmetic == "reference" else True
        if lengths is not None and not fill:
            return self._function.apply(theta, A, self.operator, lengths, xs, True)
        return self._function.apply(theta, A, self.operator, lengths, xs)

    def traceback(self, grad):
        return traceback(grad, self.traceback_rule)

    def traceback_batch(self, grad, lengths=None):
        """Extension: the same walk for a whole (B, N, M) batch on the device, one wavefront per pair,
        instead of one host walk per pair (alignment.py:165-170).  -> list of B lists of (i, j, state);
        raises IndexError if any walk leaves its matrix, like the per-pair version."""
        states, counts = _engine.get_engine().traceback(grad, lengths, self.traceback_rule)
        states, counts = states.cpu().numpy(), counts.cpu().numpy()
        if (counts < 0).any():
            raise IndexError(f"traceback walked off the matrix for pairs {np.nonzero(counts < 0)[0].tolist()}")
        return [[tuple(int(v) for v in row) for row in states[b, :counts[b]]] for b in range(len(counts))]

    def decode(self, theta, A, lengths=None, fill=True):
        """Expected alignment matrix dVt/dtheta, differentiable (nw_cuda.py:319-325).  `lengths`, `fill`: see forward()
        (the gradient that flows back through the result, Ed, is always zero outside the blocks)."""
        with torch.enable_grad():
            nll = self._forward_for_decode(theta, A, lengths, fill)
            v = torch.sum(nll)
            v_grad, _ = torch.autograd.grad(v, (theta, A), create_graph=True)
        return v_grad
