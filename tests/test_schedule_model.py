"""CPU: a Python model of the kernel's index arithmetic (deepblast_amd/csrc/sdp_kernels.hip).

It re-derives, for random shapes and both chunk lengths, the three pieces of geometry the HIP sweep relies
on, and checks their invariants exhaustively -- these are the formulas a future edit is most likely to break:

  1. the (strip, step, lane) -> cell schedule visits every cell exactly once and respects the DP
     dependencies (predecessors are older steps of the same lane / the lane above / the strip above);
  2. input staging: with blocks of K columns aligned to K, during chunk c row r only ever needs blocks
     c-q and c-q+1 (q = ceil(r/K)), the ring slot (block & 1) never collides, and the block prefetched for
     chunk c+1 overwrites a slot that chunk c no longer needs; the per-lane ring read index is
     (t - lane) mod 2K;
  3. output flush: the aligned K-column blocks written after every chunk cover every cell exactly once,
     and each element is read from the half of the 2-chunk LDS ring the kernel formula says.
"""
import numpy as np
import pytest


def ceil_div(a, b):
    return -(-a // b)


@pytest.mark.parametrize("K", [16, 32])
@pytest.mark.parametrize("N,M", [(1, 1), (64, 64), (65, 63), (130, 200), (200, 45), (70, 333)])
def test_schedule_visits_every_cell_once_in_dependency_order(N, M, K):
    nstrips = ceil_div(N, 64)
    nchunks = ceil_div(M + 63, K)
    when = -np.ones((N, M), dtype=np.int64)   # global order key: (strip, step)
    for s in range(nstrips):
        for t in range(nchunks * K):
            for lane in range(64):
                i, j = 64 * s + lane, t - lane
                if i < N and 0 <= j < M:
                    assert when[i, j] < 0
                    when[i, j] = t
    assert (when >= 0).all()
    # left / up / diagonal predecessors: same lane one step earlier; lane above one resp. two steps earlier
    i, j = np.meshgrid(np.arange(N), np.arange(M), indexing="ij")
    assert (when[:, 1:] == when[:, :-1] + 1).all()
    same_strip = (i[1:, :] % 64) != 0
    assert (when[1:, :][same_strip] == when[:-1, :][same_strip] + 1).all()
    # across strips the lane above is lane 63 of the previous strip: its column j is produced at step j+63,
    # consumed by lane 0 at step j -> the consumer strip must lag by >= 63 steps plus the publish granule K
    assert (when[:-1, :][~same_strip] == when[1:, :][~same_strip] + 63).all()


@pytest.mark.parametrize("K", [16, 32])
@pytest.mark.parametrize("M", [1, 31, 64, 100, 257, 512])
@pytest.mark.parametrize("rev", [False, True])
def test_input_ring_holds_exactly_the_needed_blocks(M, K, rev):
    nchunks = ceil_div(M + 63, K)
    order = range(nchunks - 1, -1, -1) if rev else range(nchunks)
    c_first = nchunks - 1 if rev else 0
    for r in range(64):
        q = ceil_div(r, K)
        ring = {}                                            # slot -> block index

        def write(bb):
            ring[(bb - q) & 1] = bb - q

        write(c_first)
        write(c_first + 1)
        for c in order:
            need = {(c * K + k - r) // K for k in range(K)}  # blocks touched by this row during chunk c
            assert need <= {c - q, c - q + 1}
            for blk in need:
                assert ring[blk & 1] == blk, (r, c, blk, ring)
            for k in range(K):                               # per-lane read index = column mod 2K
                col = c * K + k - r
                assert (col % (2 * K)) // K == (col // K) & 1
            nxt = c - 1 if rev else c + 1
            if 0 <= nxt < nchunks:
                bb_new = c - 1 if rev else c + 2
                victim = ring[(bb_new - q) & 1]
                nxt_need = {nxt - q, nxt - q + 1}
                assert victim not in nxt_need                # the overwritten block is dead
                write(bb_new)


@pytest.mark.parametrize("K", [16, 32])
@pytest.mark.parametrize("N,M", [(64, 64), (64, 100), (40, 7), (64, 513)])
def test_aligned_flush_covers_every_cell_once(N, M, K):
    """Reverse sweep: after chunk t0 row r flushes the K-column block starting at t0 - K*floor(r/K); element e
    comes from step offset s = (r mod K) + e: this chunk's half of the ring if s < K, else the half written by
    the previously processed chunk (t0 + K).  Model the ring literally and check values and coverage."""
    nchunks = ceil_div(M + 63, K)
    PO = 2 * K + 1
    rows = min(N, 64)
    ring = np.full((64, PO), -1, dtype=np.int64)             # holds "column index" of the cell written there
    seen = np.zeros((rows, M), dtype=np.int64)
    for c in range(nchunks - 1, -1, -1):
        t0, par = c * K, c & 1
        for k in range(K - 1, -1, -1):
            for lane in range(64):
                ring[lane, par * K + k] = t0 + k - lane      # the value E[lane, t0+k-lane] (any int stands in)
        for r in range(64):
            blk0 = t0 - K * (r // K)
            for e in range(K):
                s = (r % K) + e
                prev = s >= K
                off0 = (s - K if prev else s) + (K if prev else 0)   # fo_off0 without the row term
                dk = -K if prev else K
                idx = off0 + par * dk
                col = blk0 + e
                if 0 <= col < M and r < rows:
                    assert ring[r, idx] == col, (c, r, e)
                    seen[r, col] += 1
    assert (seen == 1).all()
