#!/usr/bin/env python
"""What any schedule of a pair's strips on W wavefronts can reach ("ramps").

A pair of S strips is S chains of L = M + 63 steps; strip s may execute its step j only after strip s - 1 has executed
step j + lag (lag = 63 + the publishing granule: the last lane of a strip trails its first lane by 63 columns).  One
wavefront executes one step of one strip at a time.  Round 3 asked whether time-slicing the strips at block granularity
(a wave alternating between its two strips) could approach max(work / W, critical path) = max(S L / W, (S - 1) lag + L).
It cannot: during [k lag, (k + 1) lag) only strips 0..k can have started, and by symmetry during the last (k + 1)-th lag
before the end only strips S - 1 - k .. S - 1 can still be unfinished, so with W waves

    T  >=  ( S L  +  2 * lag * sum_{k < W} (W - 1 - k) ) / W  =  S L / W + (W - 1) lag        (S >= 2 W - 1, L >= W lag)

whatever the assignment of strips to waves and whatever the interleaving -- and the kernel's schedule (wave w: strip w,
then w + W, ..., each run to completion) attains it.  `lower_bound` is that formula, `run_to_completion` the kernel's
schedule, `best_found` a search over time-slicing policies in a block-granular simulator: none beats run_to_completion.

    python tools/ramp_bound.py            # headline shape: 8 strips x 575 steps on 4 waves, lag 79 (forward) / 95 (backward)
"""
import heapq
import itertools
import sys


def lower_bound(S, L, lag, W):
    """Steps: work plus the waves that MUST idle while the pipeline of strips fills and drains, over W waves."""
    w = min(W, S)
    idle = sum(max(0, w - 1 - k) for k in range(w)) * lag
    return max((S * L + 2 * idle) / w, (S - 1) * lag + L)


def run_to_completion(S, L, lag, W):
    """The kernel's schedule: wave w takes strips w, w + W, ... one after the other.  -> step at which the pair is done."""
    start = [0] * S
    end = [0] * S
    for s in range(S):
        free = end[s - W] if s >= W else 0                 # the wave is busy with its previous strip until then
        ready = start[s - 1] + lag if s > 0 else 0          # the strip above is `lag` steps ahead from its start on
        start[s] = max(free, ready)
        # a strip that starts late never catches up with a predecessor that keeps running (same speed), but a
        # predecessor that started later than `lag` before us cannot be overtaken either: end >= end[s - 1] + lag
        end[s] = max(start[s] + L, (end[s - 1] + lag) if s > 0 else 0)
    return end[S - 1]


def simulate(S, L, skew, W, blk, prio, owner=None, switch_cost=0.0):
    """Block-granular preemptive schedule: whenever a wave is free it runs one block (blk steps = the publishing granule)
    of the runnable strip with the highest prio(strip, blocks done, strip it ran last); block j of strip s is runnable
    once strip s - 1 has finished the block that contains its step (j + 1) * blk - 1 + skew (skew = 63: the last lane of
    a strip meets a column 63 steps after the first).  -> steps until the last strip is done."""
    nblk = -(-L // blk)
    lagb = (blk - 1 + skew) // blk + 1   # block j needs blocks 0 .. j + lagb - 1 of the strip above
    owner = owner or (lambda s: s % W)
    done = [0] * S
    fin = [[None] * nblk for _ in range(S)]
    evs = [(0.0, w) for w in range(W)]
    heapq.heapify(evs)
    last = [None] * W
    end = 0.0
    while evs:
        now, w = heapq.heappop(evs)
        mine = [s for s in range(S) if owner(s) == w and done[s] < nblk]
        if not mine:
            continue

        def runnable(s):
            if s == 0:
                return True
            need = min(done[s] + lagb - 1, nblk - 1)
            f = fin[s - 1][need]
            return f is not None and f <= now + 1e-9
        r = [s for s in mine if runnable(s)]
        if not r:
            heapq.heappush(evs, (now + 0.125, w))
            continue
        s = max(r, key=lambda s: prio(s, done[s], last[w]))
        cost = 1.0 + (switch_cost if last[w] not in (None, s) else 0.0)
        last[w] = s
        fin[s][done[s]] = now + cost
        done[s] += 1
        end = max(end, now + cost)
        heapq.heappush(evs, (now + cost, w))
    return end * blk


def best_found(S, L, lag, W, blk=16):
    """Makespans over a family of time-slicing policies and strip-to-wave assignments (lag = 63 + blk)."""
    nblk = -(-L // blk)
    skew = lag - blk
    policies = {
        "lowest strip first (= run to completion)": lambda s, j, l: -s,
        "highest strip first": lambda s, j, l: s,
        "least progress first": lambda s, j, l: -j,
        "most remaining critical path": lambda s, j, l: (S - 1 - s) * lag / blk + (nblk - j),
        "alternate": lambda s, j, l: 0 if s == l else 1,
    }
    for h in (2, 4, 8):
        policies[f"critical path, sticky {h}"] = lambda s, j, l, h=h: (S - 1 - s) * lag / blk + (nblk - j) + (h if s == l else 0)
    owners = {"w, w+W": lambda s: s % W, "mirrored (w, 2W-1-w)": lambda s: s % W if (s // W) % 2 == 0 else W - 1 - s % W,
              "consecutive": lambda s: (s * W) // S}
    out = {}
    for (pn, pf), (on, of) in itertools.product(policies.items(), owners.items()):
        out[(pn, on)] = simulate(S, L, skew, W, blk, pf, of)
    return out


def roll_over(S, M, lag, W):
    """Strip roll-over: the bound above assumes that every strip pays its own 63-step ramp -- a strip is a
    chain of L = M + 63 steps.  It need not: the lanes of a wave can ROLL from strip w straight into strip w + W -- lane l moves to
    row 64 (w + W) + l the step after it has finished row 64 w + l; the DPP chain stays valid because lane l - 1 moved one step
    earlier -- so a wave runs ONE chain of R M + 63 "virtual" steps for its R strips, virtual step t of lane l being column
    (t - l) mod M of its strip (t - l) div M.  Simulated here step by step: wave w may execute virtual step t only when the strip
    above the one its lane 0 is in has executed its own step for that column plus `lag` (lane 63 meets a column 63 steps after lane
    0, plus the publishing granule); everything else is inside the wave.  -> (steps until the pair is done, state rows per wave).

    Closed form when nothing stalls: R M + 63 + (W - 1) lag.  Wave 0's second strip takes its boundary from wave W - 1's first,
    which runs (W - 1) lag behind wave 0: fine iff M >= W lag (512 >= 316 forward, 380 backward).  Below that the schedule does
    not merely stall, it DEADLOCKS (RuntimeError here): a wave that waits for its lane 0 waits with all its lanes, also those
    still inside the previous strip, whose last columns the next wave is waiting for -- around the ring of W waves."""
    R = [len(range(w, S, W)) for w in range(W)]            # strips per wave
    total = [r * M + 63 if r else 0 for r in R]             # virtual steps per wave
    done = [0] * W                                          # virtual steps executed
    t = 0
    while any(done[w] < total[w] for w in range(W)):
        nxt = list(done)
        for w in range(W):
            v = done[w]
            if v >= total[w]:
                continue
            k, c = divmod(v, M)                             # lane 0 is at column c of the wave's k-th strip (or past its last one)
            ok = True
            if k < R[w]:
                s = w + k * W
                if s > 0:
                    pw, pk = (s - 1) % W, (s - 1) // W      # the strip above: wave pw, its pk-th strip
                    need = min(pk * M + min(c + lag, M + 63), total[pw])  # virtual step of that wave by which column c is published (its last columns: when it is through)
                    ok = done[pw] >= need
            if ok:
                nxt[w] = v + 1
        done = nxt
        t += 1
        if t > 100 * (S * (M + 63)):
            raise RuntimeError("schedule does not progress")
    return t, max(total)


def main():
    S, M, W = 8, 512, 4
    L = M + 63
    for name, lag in (("forward (publishes per 16-step block)", 79), ("backward (publishes per 32-step chunk)", 95)):
        lb, rtc = lower_bound(S, L, lag, W), run_to_completion(S, L, lag, W)
        print(f"{name}: lag {lag}: work/W = {S * L / W:.0f}, critical path = {(S - 1) * lag + L}, bound with the ramps = {lb:.0f}, "
              f"kernel's schedule = {rtc} steps")
        res = best_found(S, L, lag, W, blk=lag - 63)
        best = min(res.values())
        print(f"   block-granular time-slicing, {len(res)} policy x assignment combinations: best {best:.0f} steps "
              f"({min(res, key=res.get)}), worst {max(res.values()):.0f}")
        ro, rows = roll_over(S, M, lag, W)
        print(f"   strip roll-over (one ramp per WAVE, not per strip): {ro} steps ({100.0 * (ro - rtc) / rtc:+.1f} %), state rows per wave {rows} "
              f"instead of {S // W * L} (skew padding {rows / (S // W * M):.4f} instead of {L / M:.4f})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
