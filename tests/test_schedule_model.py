"""CPU: a Python model of the kernel's index arithmetic (deepblast_amd/csrc/sdp_kernels.hip).

It re-derives, for random shapes and both chunk lengths, the three pieces of geometry the HIP sweep relies
on, and checks their invariants exhaustively -- these are the formulas a future edit is most likely to break:

  1. the (strip, step, lane) -> cell schedule visits every cell exactly once and respects the DP
     dependencies (predecessors are older steps of the same lane / the lane above / the strip above);
  2. input staging: row r's K-column blocks start at columns K*j - (r mod 4); during chunk c the row only
     needs blocks c-q and c-q+1 (q = ceil(4*floor(r/4)/K)); the ring position of column j is
     (j + r + 4*pi(r mod 8)) mod 2K, so that every dwordx4 lands on an aligned 16-byte slot, lane l reads step
     t at (t + 4*pi(l mod 8)) mod 2K, and the block prefetched for chunk c+1 only overwrites dead data;
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


def ring_pi(x):
    return ((x & 1) << 2) | (x >> 1)


@pytest.mark.parametrize("K", [16, 32])
@pytest.mark.parametrize("M", [1, 31, 64, 100, 257, 512])
@pytest.mark.parametrize("rev", [False, True])
@pytest.mark.parametrize("lines", [False, True, "shifted"])
def test_input_ring_holds_exactly_the_needed_blocks(M, K, rev, lines):
    """Model the staged-input ring literally: lanes load dwordx4 groups of row r's blocks (which start at
    columns K*j - (r mod 4), or at K*j in the line-aligned variant), write each group to one aligned 16-byte slot
    (four dword writes in the line-aligned variant), and lane l reads the K values of a
    chunk as K/4 aligned 16-byte groups at position (t + 4*pi(l mod 8)) mod 2K.  Check that every value read
    for a real column is that column, that writes only overwrite dead data, and that all slots are aligned."""
    if lines and K != 32:
        pytest.skip("the line-aligned staging exists for K = 32 only (its row assignment is written for 8 loads of 8 rows)")
    nchunks = ceil_div(M + 63, K)
    RING = 2 * K
    LPR, RPL, NLD = K // 4, 64 // (K // 4), K // 4
    # "shifted": the line-aligned variant for a plane / row pitch that is not aligned to K floats: row r's blocks move
    # left by delta_r = (r M + beta) mod K so that they start on K-float boundaries of MEMORY
    beta = 5 if lines == "shifted" else 0
    pitch_mod = (M % K) if lines == "shifted" else 0
    order = range(nchunks - 1, -1, -1) if rev else range(nchunks)
    c_first = nchunks - 1 if rev else 0
    ring = np.full((64, RING), -10**9, dtype=np.int64)      # holds the column index stored in each position

    def write_block(bb):
        flip = (bb & 1) * K
        for i in range(NLD):
            for lane in range(64):
                r4, cg = lane // LPR, lane % LPR
                r = i * RPL + r4
                if lines:   # blocks start at multiples of K (whole lines)
                    # load instruction i takes the rows with r mod 4 == i mod 4 of one half of the strip (round 4): every
                    # lane of an instruction then cuts its group of four columns into the same aligned pieces
                    r = (i & 3) + 32 * (i >> 2) + 4 * r4
                    delta = (r * pitch_mod + beta) % K
                    q = (r - delta + K - 1) // K
                    assert q >= 0 and K * q + delta >= r and K * q + delta < r + K   # the two live blocks cover the window
                    col0 = K * (bb - q) - delta + 4 * cg
                    assert (beta + r * pitch_mod + K * (bb - q) - delta) % K == 0    # block starts on a K-float boundary
                    ws = []
                    for j in range(4):
                        w = ((4 * cg + j - delta + RING + K * (q & 1) + r + 4 * ring_pi(r & 7)) & (RING - 1)) ^ flip
                        assert w == (col0 + j + r + 4 * ring_pi(r & 7)) % RING
                        ring[r, w] = col0 + j
                        ws.append(w)
                    if lines is True:   # aligned pitch and plane: 16 | 8 + 8 | 4 + 8 + 4 byte writes, every piece aligned to its size
                        rho = i & 3
                        assert ws[0] % 4 == rho
                        pieces = {0: [(0, 4)], 2: [(0, 2), (2, 2)], 1: [(0, 1), (1, 2), (3, 1)], 3: [(0, 1), (1, 2), (3, 1)]}[rho]
                        for j0, n in pieces:
                            assert ws[j0] % n == 0 and all(ws[j0 + e] == ws[j0] + e for e in range(n)), (r, ws, j0, n)
                    continue
                q = ceil_div(r & ~3, K)
                col0 = K * (bb - q) - (r & 3) + 4 * cg          # li_voff without the row term
                w = ((4 * cg + K * (q & 1) + (r & ~3) + 4 * ring_pi(r & 7)) & (RING - 1)) ^ flip
                assert w % 4 == 0                               # one aligned ds_write_b128
                assert w == (col0 + r + 4 * ring_pi(r & 7)) % RING
                ring[r, w:w + 4] = np.arange(col0, col0 + 4)

    write_block(c_first)
    write_block(c_first + 1)
    for c in order:
        t0 = c * K
        for lane in range(64):
            pr = (t0 & (RING - 1)) + 4 * ring_pi(lane & 7)
            for g in range(K // 4):
                idx = (pr + 4 * g) & (RING - 1)
                assert idx % 4 == 0                             # one aligned ds_read_b128
                for e in range(4):
                    col = t0 + 4 * g + e - lane
                    if 0 <= col < M:
                        assert ring[lane, idx + e] == col, (c, lane, g, e)
        nxt = c - 1 if rev else c + 1
        if 0 <= nxt < nchunks:
            write_block(c - 1 if rev else c + 2)
            # the block just written must not have destroyed anything chunk `nxt` reads (checked when it runs)


@pytest.mark.parametrize("K", [16, 32])
@pytest.mark.parametrize("beta", [0, 5])
@pytest.mark.parametrize("N,M", [(64, 64), (64, 100), (40, 7), (64, 513), (64, 516), (64, 500), (33, 129)])
def test_aligned_flush_covers_every_cell_once(N, M, K, beta):
    """Reverse sweep: after chunk t0 row r flushes the K-element block starting at column t0 - D_r, D_r = r - rho_r,
    rho_r = (r (1 - M) - beta) mod K, so that the block starts on a multiple of K floats in MEMORY (beta = misalignment
    of the plane); element e comes from step offset s = rho_r + e: this chunk's half of the ring if s < K, else the half
    written by the previously processed chunk (t0 + K).  Model the ring literally and check values, coverage and the
    alignment of every block.  (M a multiple of K, beta = 0: rho_r = r mod K, the column-aligned blocks of round 1.)"""
    nchunks = ceil_div(M + 63, K)
    PO = 2 * K + 1
    rows = min(N, 64)
    ring = np.full((64, PO), -1, dtype=np.int64)             # holds "column index" of the cell written there
    seen = np.zeros((rows, M), dtype=np.int64)
    need_tail = beta != 0 or M % K != 0
    for c in range(nchunks - 1, -2, -1):   # c = -1: the tail flush (t0 = -K) for rows whose blocks start right of t0
        t0, par = c * K, c & 1
        if c >= 0:
            for k in range(K - 1, -1, -1):
                for lane in range(64):
                    ring[lane, par * K + k] = t0 + k - lane      # the value E[lane, t0+k-lane] (any int stands in)
        elif not need_tail:
            continue
        for r in range(64):
            rho = (r * (1 - M) - beta) % K
            d = r - rho
            blk0 = t0 - d
            assert (beta + r * M + blk0) % K == 0                    # starts on a K-float boundary in memory
            if M % K == 0 and beta == 0:
                assert rho == r % K and d == K * (r // K)
            for e in range(K):
                s = rho + e
                prev = s >= K
                off0 = (s - K if prev else s) + (K if prev else 0)   # fo_off0 without the row term
                dk = -K if prev else K
                idx = off0 + par * dk
                assert idx == (s + par * K) % (2 * K)                   # (round 6: the form the general-pitch K = 32 builds compute on the spot)
                col = blk0 + e
                if 0 <= col < M and r < rows:
                    assert ring[r, idx] == col, (c, r, e)
                    seen[r, col] += 1
    assert (seen == 1).all()


@pytest.mark.parametrize("N,M", [(64, 64), (64, 96), (64, 100), (40, 7), (64, 512), (64, 513), (64, 500), (33, 129), (64, 33)])
def test_wide_flush_of_a_short_chunk_sweep(N, M):
    """Round 6, adjoint backward sweep (sdp_kernels.hip: FLUSH2 with KF = 32 at K = 16): the sweep runs 16-step chunks, but a
    row's outputs leave as 32-column blocks -- whole 128-byte lines -- out of a ring of 64 steps per row, after every SECOND
    chunk (the one whose t0 is a multiple of 32), at the top of the next iteration.  Element e of row r's block comes from step
    offset s = (r mod 32) + e of the ring, counted from the block's own half (this super-chunk) into the other half (the one
    processed before); the pair that would straddle ring positions 63 | 0 is read from position -1 of the row, where the chunk
    that holds a block's last step leaves a copy of it.  Model ring, cadence and position -1 literally; check values and that
    every cell leaves exactly once."""
    K, KF, PO = 16, 32, 65
    nchunks = ceil_div(M + 63, K)
    rows = min(N, 64)
    ring = np.full((64, PO + 1), -10**6, dtype=np.int64)     # column 0 of this array is "position -1" of the row
    pos = lambda p_: p_ + 1
    seen = np.zeros((rows, M), dtype=np.int64)
    pf_t0 = 0
    for ci in range(nchunks + 1):                              # one extra iteration for the last chunk's flush
        if (pf_t0 & (KF - 1)) == 0 and ci > 0:                 # (ci = 0: the kernel's flush runs with every store masked)
            t0, par = pf_t0, (pf_t0 // KF) & 1
            for r in range(64):
                rho = r & (KF - 1)
                d = r - rho
                for e in range(0, KF, 2):                      # the kernel reads pairs (8-byte LDS reads)
                    sb = rho + e
                    base = sb + ((KF if sb < KF - 1 else -KF) if par else 0)
                    for j in range(2):
                        col = t0 - d + e + j
                        if 0 <= col < M and r < rows:
                            assert ring[r, pos(base + j)] == col, (ci, r, e, j, par)
                            seen[r, col] += 1
        if ci < nchunks:
            c = nchunks - 1 - ci
            t0 = c * K
            q = t0 & (2 * KF - 1)
            for k in range(K - 1, -1, -1):
                for lane in range(64):
                    ring[lane, pos(q + k)] = t0 + k - lane     # the value Ed[lane, t0 + k - lane] (any int stands in)
            if ((t0 + K) & (KF - 1)) == 0:                     # the chunk holds the last step of a flush block
                ring[:, pos(-1)] = ring[:, pos(q + K - 1)]
            pf_t0 = t0
    assert (seen == 1).all()


def test_no_time_slicing_of_strips_beats_run_to_completion():
    """DESIGN.md section 4 ("ramps"): with W waves per pair the pipeline of strips has to fill and drain -- during the
    k-th lag after the start only strips 0..k can have begun, and symmetrically before the end -- so
    T >= S L / W + (W - 1) lag steps for ANY assignment / interleaving of strips, and the kernel's schedule (wave w runs
    strips w, w + W, ... to completion) attains it: 1387 steps at the headline shape, not the max(work, critical path)
    = 1150 that time-slicing at block granularity was hoped to approach (tools/ramp_bound.py)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import ramp_bound as rb
    for S, M, W, lag in ((8, 512, 4, 79), (8, 512, 4, 95), (16, 1024, 4, 79), (8, 512, 2, 79), (12, 300, 4, 79)):
        L = M + 63
        lb, rtc = rb.lower_bound(S, L, lag, W), rb.run_to_completion(S, L, lag, W)
        assert rtc == lb == S * L // W + (W - 1) * lag, (S, M, W, lag, lb, rtc)
    assert rb.run_to_completion(8, 575, 79, 4) == 1387
    res = rb.best_found(8, 575, 79, 4, blk=16)
    # (the simulator runs whole 16-step blocks: its run-to-completion figure is the formula's rounded up to blocks)
    assert min(res.values()) == res[("lowest strip first (= run to completion)", "w, w+W")] >= 1387
    assert min(res.values()) <= 1387 + 16


def test_strip_roll_over_pays_the_ramp_once_per_wave():
    """VERDICT r5 item 4.  The "ramp bound" T >= S L / W + (W - 1) lag takes a strip for a chain of L = M + 63 steps: every strip
    pays its own 63-step ramp.  A wave whose lanes ROLL from strip w into strip w + W (lane l changes rows the step after it has
    finished its row; lane l - 1 changed one step earlier, so the DPP chain stays valid) runs one chain of R M + 63 steps for
    its R strips: 2 x 512 + 63 + 3 x 79 = 1324 steps against 1387 at the headline shape (-4.5 %), 1372 against 1435 in the
    backward sweep, and the skew padding of the state falls from 1.123 to 1.0615.  tools/ramp_bound.py simulates the schedule
    step by step (the hand-off between strips with its lag; a wave's own lanes need nothing from outside); it stalls only
    when M < W lag -- wave 0's second strip then runs into wave W - 1's first."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import ramp_bound as rb
    assert rb.roll_over(8, 512, 79, 4) == (1324, 1087)
    assert rb.roll_over(8, 512, 95, 4) == (1372, 1087)
    for S, M, W, lag in ((8, 512, 4, 79), (16, 1024, 4, 79), (16, 1024, 4, 95), (8, 512, 2, 79), (12, 400, 4, 95), (6, 512, 4, 79), (9, 640, 4, 79)):
        R = -(-S // W)
        steps, rows = rb.roll_over(S, M, lag, W)
        assert M >= W * lag
        # every wave with R strips ends at R M + 63 + w lag; the last strip belongs to wave (S - 1) % W
        last = (S - 1) % W
        assert steps == len(range(last, S, W)) * M + 63 + last * lag or steps == R * M + 63 + ((S - 1 - (R - 1) * W)) * lag, (S, M, W, lag, steps)
        assert rows == R * M + 63
        assert steps < rb.run_to_completion(S, M + 63, lag, W)
    # short rows: M < W lag is not a stall but a DEADLOCK -- wave 0 cannot take the first step of its second strip before wave
    # W - 1 has published column 0 of its first one (virtual step `lag` of a wave that runs (W - 1) lag behind), and a wave that
    # waits waits with all its lanes, also those still inside the previous strip, whose last columns wave 1 is waiting for, and
    # so on around the ring.  A build with roll-over therefore needs M >= W lag (the host would have to choose per launch).
    import pytest
    with pytest.raises(RuntimeError):
        rb.roll_over(8, 256, 79, 4)
    assert rb.roll_over(8, 316, 79, 4)[0] == 2 * 316 + 63 + 3 * 79
