"""CPU restatement of the reference's score construction (TEST INFRASTRUCTURE ONLY; see oracle/sdp_oracle.c).

deepblast/alignment.py:122-123 (NeuralAligner.forward) and :134-135 (.score):
    theta = F.softplus(torch.einsum('bid,bjd->bij', zx, zy))
    A     = F.logsigmoid(torch.einsum('bid,bjd->bij', gx, gy))
torch computes both lines in fp32; the inner products here are accumulated in float64 and rounded once, so that the
oracle is the value every fp32 summation order approximates.  softplus follows torch's definition (beta 1, threshold
20).  Pinned by tests/golden/g11_scores.npz, which oracle/gen_golden_scores.py wrote from the real torch ops.
"""
import numpy as np


def softplus(s):
    s = np.asarray(s, np.float64)
    return np.where(s > 20.0, s, np.maximum(s, 0.0) + np.log1p(np.exp(-np.abs(s))))


def logsigmoid(s):
    s = np.asarray(s, np.float64)
    return np.minimum(s, 0.0) - np.log1p(np.exp(-np.abs(s)))


def scores(zx, zy, gx, gy):
    """-> (theta, A) float32 (B,N,M)."""
    st = np.einsum("bid,bjd->bij", np.asarray(zx, np.float64), np.asarray(zy, np.float64))
    sa = np.einsum("bid,bjd->bij", np.asarray(gx, np.float64), np.asarray(gy, np.float64))
    return softplus(st).astype(np.float32), logsigmoid(sa).astype(np.float32)
