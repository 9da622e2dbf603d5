"""Batch-sharded alignment across the GPUs of one node (one process per GPU, RCCL over xGMI).

The DP is embarrassingly parallel over pairs (the reference loops `for b in range(B)`,
deepblast/nw.py:110), so the data path needs NO collective: every rank aligns its own pairs.
The only exchange is collecting results afterwards -- one all-gather of Vt (B floats) and,
on request, of the expected-alignment matrices E.  `torch.distributed` backend "nccl" is RCCL
on ROCm; on CPU test boxes the same code runs over "gloo".
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(B, world, rank):
    """Contiguous [lo, hi) slice of a batch of B pairs owned by `rank` (sizes differ by <= 1)."""
    base, rem = divmod(B, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def balanced_assignment(work, world):
    """Equal-count, work-balanced assignment of pairs to ranks for variable-length batches.

    Pairs are sorted by work (n_b * m_b) descending and dealt in a snake (0..G-1, G-1..0, ...),
    which keeps per-rank counts within one of each other and per-rank work within one pair of the
    mean.  Returns (order, inverse): rank r takes order[r::world]-style slices via `take(r)`;
    `inverse` restores the original batch order after a gather.
    """
    work = np.asarray(work, dtype=np.int64)
    B = work.shape[0]
    by_work = np.argsort(-work, kind="stable")
    buckets = [[] for _ in range(world)]
    for pos, idx in enumerate(by_work):
        rnd, k = divmod(pos, world)
        buckets[k if rnd % 2 == 0 else world - 1 - k].append(int(idx))
    order = np.array([i for bkt in buckets for i in bkt], dtype=np.int64)
    counts = np.array([len(bkt) for bkt in buckets], dtype=np.int64)
    inverse = np.empty(B, dtype=np.int64)
    inverse[order] = np.arange(B)
    return order, counts, inverse


def _all_gather_cat(x, group, async_op=False):
    """All-gather equally shaped tensors along dim 0 with one collective.

    async_op=True returns (out, work): the collective is enqueued on the backend's own stream and `work.wait()`
    makes the current stream wait for it, so that it can overlap with kernels launched in between."""
    world = dist.get_world_size(group)
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    try:
        work = dist.all_gather_into_tensor(out, x.contiguous(), group=group, async_op=async_op)
    except (RuntimeError, NotImplementedError):  # backends without the flat variant
        parts = list(out.chunk(world, dim=0))
        work = dist.all_gather(parts, x.contiguous(), group=group, async_op=async_op)
    return (out, work) if async_op else out


class ShardedAligner:
    """Align this rank's shard and collect results from all ranks.

    decoder : NeedlemanWunschDecoder / SmithWatermanDecoder
    gather  : "vt" (default) collect terminal scores only; "e" also collect E; "none" nothing.
              Gathering E moves (G-1) x B/G x N x M x 4 bytes INTO every GPU over xGMI -- at
              B/G=256, N=M=512 that is 1.9 GB per rank and takes several times longer than
              computing it, so it is opt-in.
    """

    def __init__(self, decoder, group=None, gather="vt"):
        if gather not in ("vt", "e", "none"):
            raise ValueError("gather must be 'vt', 'e' or 'none'")
        self.decoder = decoder
        self.group = group
        self.gather = gather
        self._ones = None

    def align(self, theta, A, lengths=None):
        """theta, A: this rank's (B_local, N, M) shard.  Every rank must hold the same B_local
        (pad the last shard) -- the all-gather is a single fixed-size collective.

        -> dict(Vt_local, E_local, Vt (world*B_local,) or None, E (world*B_local,N,M) or None)."""
        theta = theta.detach().requires_grad_(True)
        Vt = self.decoder(theta, A, lengths) if lengths is not None else self.decoder(theta, A)
        # dVt.sum()/dtheta with the cotangent handed over directly: no reduction kernel and no expand/copy of
        # its gradient on the way to the backward sweep
        if self._ones is None or self._ones.shape != Vt.shape or self._ones.device != Vt.device:
            self._ones = torch.ones_like(Vt)
        gathering = self.gather != "none" and dist.is_available() and dist.is_initialized()
        pending = None
        if gathering:
            # the scores are final after the forward sweep: their all-gather (RCCL's own stream) runs while the
            # backward sweep computes E
            pending = _all_gather_cat(Vt.detach(), self.group, async_op=True)
        (E,) = torch.autograd.grad(Vt, theta, grad_outputs=self._ones)
        out = {"Vt_local": Vt.detach(), "E_local": E, "Vt": None, "E": None}
        if gathering:
            vt_all, work = pending
            if work is not None:
                work.wait()
            out["Vt"] = vt_all
            if self.gather == "e":
                out["E"] = _all_gather_cat(E, self.group)
        return out
