"""Batch-sharded alignment across the GPUs of one node (one process per GPU, RCCL over xGMI).

The DP is embarrassingly parallel over pairs (the reference loops `for b in range(B)`,
deepblast/nw.py:110), so the data path needs NO collective: every rank aligns its own pairs.
The only exchange is collecting results afterwards -- one all-gather of Vt (B floats) and,
on request, of the expected-alignment matrices E.  `torch.distributed` backend "nccl" is RCCL
on ROCm; on CPU test boxes the same code runs over "gloo".

Two ways to cut a batch:
  * `shard_bounds`  -- contiguous B/G slices (equal shapes: BASELINE.json configs[4]);
  * `BalancedPlan`  -- for padded batches with per-pair lengths: pairs are dealt to ranks by work
    n_b*m_b (longest first, snake order) so that every rank gets the same number of pairs and
    nearly the same number of DP cells, and the gathered results are put back into batch order.
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
    mean.  Returns (order, counts, inverse): rank r owns order[sum(counts[:r]) : sum(counts[:r+1])];
    `inverse` restores the original batch order after a gather of the per-rank results.
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


class BalancedPlan:
    """Which pairs of a variable-length batch each rank aligns, and how to restore batch order.

    Built from the global (B, 2) lengths, which every rank knows (it is the collate side-channel,
    deepblast_amd/batching.py); deterministic, so every rank computes the same plan without talking.
    Every rank aligns `per_rank` pairs: ranks that were dealt one pair fewer (B not a multiple of the
    world size) pad with a 1x1 dummy pair, so the gather stays one fixed-size collective.
    """

    def __init__(self, lengths, world):
        lengths = np.asarray(lengths.cpu() if isinstance(lengths, torch.Tensor) else lengths, dtype=np.int64)
        if lengths.ndim != 2 or lengths.shape[1] != 2:
            raise ValueError(f"lengths must be (B, 2), got {lengths.shape}")
        self.B, self.world = int(lengths.shape[0]), int(world)
        self.order, self.counts, _ = balanced_assignment(lengths[:, 0] * lengths[:, 1], world)
        self.per_rank = int(self.counts.max()) if self.B else 0
        self.offsets = np.concatenate([[0], np.cumsum(self.counts)])
        # position of pair b in the gathered (world * per_rank) result
        self.gathered_pos = np.empty(self.B, dtype=np.int64)
        for r in range(world):
            mine = self.order[self.offsets[r]:self.offsets[r + 1]]
            self.gathered_pos[mine] = r * self.per_rank + np.arange(len(mine))
        self.work_per_rank = np.array([int((lengths[self.indices(r), 0] * lengths[self.indices(r), 1]).sum())
                                       for r in range(world)], dtype=np.int64)

    def indices(self, rank):
        """Global batch indices of the pairs `rank` aligns (longest first)."""
        return self.order[self.offsets[rank]:self.offsets[rank + 1]]

    def restore(self, gathered):
        """(world * per_rank, ...) gathered results -> (B, ...) in the original batch order."""
        idx = torch.as_tensor(self.gathered_pos, device=gathered.device)
        return gathered.index_select(0, idx)


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


class PendingGather:
    """Result of an asynchronous gather: `.wait()` makes the current stream wait for the collective and returns
    the gathered tensor (in batch order when a plan was given)."""

    def __init__(self, out, work, plan=None):
        self._out, self._work, self._plan = out, work, plan

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None
        return self._plan.restore(self._out) if self._plan is not None else self._out


class ShardedAligner:
    """Align this rank's shard and collect results from all ranks.

    decoder : NeedlemanWunschDecoder / SmithWatermanDecoder
    gather  : "vt" (default) collect terminal scores only; "e" also collect E; "paths" also collect the
              tracebacks of E -- the device walk (sdp_traceback_i32) on each rank's own E, one int32 per step
              (i, j, state packed), (N+M+2) x 4 bytes per pair instead of N x M x 4: 130x less than E at 512 x 512,
              so the gather all but disappears ("compact tracebacks"); "none" nothing.
              Gathering E moves (G-1) x B/G x N x M x 4 bytes INTO every GPU over xGMI -- at
              B/G=256, N=M=512 that is 1.9 GB per rank and takes several times longer than
              computing it, so it is opt-in.
    async_e : with gather="e", return `out["E"]` as a PendingGather instead of waiting: the collective runs on
              RCCL's stream while the caller launches the next batch's sweeps (the gather is several times
              longer than the compute, so the compute disappears under it).
    e_chunks: with gather="e" (and no per-pair lengths): the backward sweep is launched in this many pieces of
              B_local / e_chunks pairs, and each piece's all-gather is issued as soon as the piece is enqueued -- on
              the backend's own stream, so it runs under the sweep of the next piece ("gather in chunks
              overlapped with the backward kernel").  Every piece lands directly in its place in the (B, N, M) result.
              1 = one collective after the whole sweep.  `out["e_overlap"]` says which happened ("chunked" / "none").
    """

    def __init__(self, decoder, group=None, gather="vt", async_e=False, e_chunks=1, idiom="grad"):
        if gather not in ("vt", "e", "paths", "none"):
            raise ValueError("gather must be 'vt', 'e', 'paths' or 'none'")
        if int(e_chunks) < 1:
            raise ValueError("e_chunks must be >= 1")
        if idiom not in ("grad", "sum_backward"):
            raise ValueError("idiom must be 'grad' or 'sum_backward'")
        # how E = dVt.sum()/dtheta is asked for: "grad" hands autograd a cached (B,) cotangent of ones
        # (torch.autograd.grad: no reduction kernel, no .grad accumulation); "sum_backward" is the reference user's own
        # two lines, `Vt = dec(theta, A); Vt.sum().backward()` (the idiom the headline metric is defined on)
        self.idiom = idiom
        self.decoder = decoder
        self.group = group
        self.gather = gather
        self.async_e = async_e
        self.e_chunks = int(e_chunks)
        self._ones = None

    def _world(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def align(self, theta, A, lengths=None, plan=None):
        """theta, A: this rank's (B_local, N, M) shard.  Every rank must hold the same B_local -- the all-gather
        is a single fixed-size collective (use `pad_shard` or a BalancedPlan, which pads by itself); with e_chunks > 1
        the number of collectives follows from B_local too, so unequal shards would not only gather garbage but hang.

        lengths : (B_local, 2) per-pair sizes of a padded shard (lengths-aware decode), or None.
        plan    : a BalancedPlan; theta/A/lengths are then this rank's `plan.indices(rank)` pairs (in that order),
                  and the gathered results come back in the ORIGINAL batch order.

        -> dict(Vt_local, E_local, Vt (B,) or None, E (B,N,M) | PendingGather | None,
                paths (gather="paths"): (states (B, N+M+2, 3) int32, counts (B,) int32) as Decoder.traceback_batch
                reads them -- pair b's walk is states[b, :counts[b]]; counts < 0: the walk left the matrix).
        With gather="paths" AND per-pair lengths, E_local is produced without its zero fill: outside each pair's
        [n_b, m_b] block it holds uninitialised memory (possibly NaN), because the device walk -- the only consumer in
        that mode -- masks by the same lengths.  Reduce / log / plot it only inside the blocks, or use another gather mode."""
        n_real = theta.shape[0]
        if plan is not None:
            if lengths is None:
                raise ValueError("a BalancedPlan needs the per-pair lengths of this rank's pairs")
            theta, A, lengths = pad_shard(theta, A, lengths, plan.per_rank)
        gathering = self.gather != "none" and dist.is_available() and dist.is_initialized()
        # (float64 tensors take the unchunked path: sdp_backward_range_f32 sweeps the float32 states only -- decided here,
        #  before any collective has been issued)
        if (gathering and self.gather == "e" and self.e_chunks > 1 and lengths is None and theta.shape[0] >= self.e_chunks
                and theta.dtype == torch.float32):
            return self._align_chunked_e(theta.detach(), A.detach(), n_real)
        theta = theta.detach().requires_grad_(True)
        # A takes no part in what is asked for here (E = dVt.sum()/dtheta): detached, so that neither idiom leaves anything
        # in the caller's graph -- with A attached, `Vt.sum().backward()` would accumulate the reference's pass-through
        # "gradient" (A itself, nw.py:355) into A.grad and run the backward of whatever produced A, on every align()
        A = A.detach()
        # gather="paths": the walks are all that leaves this rank, and the device walk masks by the lengths -- E outside the
        # pairs' blocks is then never read and not zero-filled (E_local in the result holds unspecified values there)
        lean = lengths is not None and self.gather == "paths" and gathering
        Vt = (self.decoder(theta, A, lengths, **({"fill": False} if lean else {})) if lengths is not None else self.decoder(theta, A))
        # idiom "grad": dVt.sum()/dtheta with the cotangent handed over directly -- no reduction kernel, no fill
        if self.idiom == "grad" and (self._ones is None or self._ones.shape != Vt.shape or self._ones.device != Vt.device):
            self._ones = torch.ones_like(Vt)
        pending = None
        if gathering:
            # the scores are final after the forward sweep: their all-gather (RCCL's own stream) runs while the
            # backward sweep computes E
            pending = _all_gather_cat(Vt.detach(), self.group, async_op=True)
        if self.idiom == "sum_backward":
            Vt.sum().backward(inputs=[theta])   # only the local theta leaf receives a gradient
            E = theta.grad
        else:
            (E,) = torch.autograd.grad(Vt, theta, grad_outputs=self._ones)
        out = {"Vt_local": Vt.detach()[:n_real], "E_local": E[:n_real], "Vt": None, "E": None, "paths": None, "e_overlap": "none"}
        if gathering:
            if self.gather == "paths":
                from . import _engine
                # the walk rule is the decoder's (Decoder(operator, traceback_rule=...)): the gathered walks must be the
                # ones decoder.traceback_batch would return for the same matrices
                states, counts = _engine.get_engine().traceback(E, lengths, rule=getattr(self.decoder, "traceback_rule", "cpu"))
                packed = pack_paths(states, counts, theta.shape[2])
                got = PendingGather(*_all_gather_cat(packed, self.group, async_op=True), plan=plan)
            out["Vt"] = PendingGather(*pending, plan=plan).wait()
            if self.gather == "paths":
                out["paths"] = unpack_paths(got.wait())
            if self.gather == "e":
                e_pending = PendingGather(*_all_gather_cat(E, self.group, async_op=True), plan=plan)
                out["E"] = e_pending if self.async_e else e_pending.wait()
        return out


    def _align_chunked_e(self, theta, A, n_real):
        """gather="e" with e_chunks > 1: backward sweep and E gather in pieces.  The full result is laid out rank-major
        like the one-collective gather's ((world * B_local, N, M)); this rank's sweep writes its pieces straight into
        its own rows of that tensor, and each piece's all-gather fills the same rows of the other ranks' blocks (the
        collective's output list are views of the result: no staging copy on our side)."""
        from . import _dp, _engine
        from .sw import SmithWatermanDecoder
        # this path calls the engine directly, so it repeats what decoder.forward checks on the way in (operator, dtype,
        # A's shape and device: the kernels take B, N, M from theta and read A through a raw pointer)
        _dp._validate(theta, A, getattr(self.decoder, "operator", "softmax"), False)
        eng = _engine.get_engine()
        variant = _engine.SW if isinstance(self.decoder, SmithWatermanDecoder) else _engine.NW
        world, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        Bl, N, M = theta.shape
        # the same two sweeps decoder(theta, A) + autograd.grad launch, in the decoder's arithmetic (sdp_backward_range_f32
        # has a reference-rounding branch)
        xs = _engine.REF if getattr(self.decoder, "arithmetic", "fast") == "reference" else False
        Vt, state = eng.forward(theta, A, variant, exact_state=xs)
        vt_pending = _all_gather_cat(Vt, self.group, async_op=True)
        if self._ones is None or self._ones.shape != Vt.shape or self._ones.device != Vt.device:
            self._ones = torch.ones_like(Vt)
        full = torch.empty((world * Bl, N, M), dtype=torch.float32, device=theta.device)
        mine = full[rank * Bl:(rank + 1) * Bl]
        bounds = [shard_bounds(Bl, self.e_chunks, k) for k in range(self.e_chunks)]
        works = []
        for lo, hi in bounds:
            eng.backward(self._ones, state, (Bl, N, M), variant, exact_state=xs, pair_range=(lo, hi), out=mine)
            outs = [full[r * Bl + lo:r * Bl + hi] for r in range(world)]
            works.append(dist.all_gather(outs, mine[lo:hi], group=self.group, async_op=True))
        e_pending = PendingGather(full, _Works(works))
        out = {"Vt_local": Vt.detach()[:n_real], "E_local": mine[:n_real], "paths": None, "e_overlap": "chunked",
               "Vt": PendingGather(*vt_pending).wait(), "E": e_pending if self.async_e else e_pending.wait()}
        return out


class _Works:
    """Several collectives' handles behind one .wait()."""

    def __init__(self, works):
        self._works = works

    def wait(self):
        for w in self._works:
            if w is not None:
                w.wait()


# One traceback step in one int32 for the gather: state in bits 0-1, j above it in ceil(log2 M) bits, i in the rest
# (31 - 2 - 12 = 17 bits at M = 2048, and N * M <= 2^28 keeps N within them).  Two header columns carry the
# pair's step count (negative: the walk left the matrix) and the width of the j field.
def _j_bits(M):
    return max(1, int(M - 1).bit_length())


def pack_paths(states, counts, M):
    """(B, cap, 3) int32 triples + (B,) counts -> (B, cap + 2) int32."""
    jb = _j_bits(M)
    s = states.to(torch.int64)
    valid = torch.arange(s.shape[1], device=s.device)[None, :] < counts.to(torch.int64)[:, None]
    s = torch.where(valid[..., None], s, torch.zeros_like(s))   # rows past the count are uninitialised memory
    if s.numel() and int(s[..., 0].max()) >= 1 << (31 - jb - 2):
        raise ValueError("traceback row index does not fit the packed word")
    if s.numel() and int(s[..., :2].min()) < 0:
        # the CPU-rule walk reproduces Python's negative-index wrap (nw.py:423) on matrices that are not alignment
        # matrices; such coordinates have no place in the packed word
        raise ValueError("a traceback walk wrapped around an edge of its matrix (negative coordinates): gather='paths' "
                         "cannot carry it -- gather E instead, or use traceback_rule='cuda'")
    word = (s[..., 0] << (jb + 2)) | (s[..., 1] << 2) | s[..., 2]
    head = torch.stack([counts.to(torch.int64), torch.full_like(counts, jb, dtype=torch.int64)], dim=1)
    return torch.cat([head, word], dim=1).to(torch.int32)


def unpack_paths(packed):
    """Inverse of pack_paths -> (states (B, cap, 3) int32, counts (B,) int32)."""
    counts = packed[:, 0].contiguous()
    jb = int(packed[0, 1]) if packed.shape[0] else 1
    word = packed[:, 2:].to(torch.int64)
    states = torch.stack([word >> (jb + 2), (word >> 2) & ((1 << jb) - 1), word & 3], dim=2).to(torch.int32)
    return states, counts


def pad_shard(theta, A, lengths, count):
    """Pad a shard to `count` pairs with 1x1 dummy pairs (zeros, lengths (1,1)) so that every rank launches and
    gathers the same shapes.  -> (theta, A, lengths as an int32 tensor on theta's device)."""
    B = theta.shape[0]
    lengths = torch.as_tensor(lengths, dtype=torch.int32, device=theta.device)
    if B > count:
        raise ValueError(f"shard has {B} pairs, more than the {count} it is padded to")
    if B == count:
        return theta, A, lengths
    extra = count - B
    zeros = theta.new_zeros((extra,) + tuple(theta.shape[1:]))
    ones = torch.ones((extra, 2), dtype=torch.int32, device=theta.device)
    return torch.cat([theta, zeros]), torch.cat([A, zeros]), torch.cat([lengths, ones])
