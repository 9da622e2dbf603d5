"""CPU: the N>1 path (batch sharding + all-gather) with world_size 2 over gloo.  The kernels are
replaced by the oracle-backed fake engine; what is under test is the sharding, the collective
and the restored ordering (deepblast_amd/distributed.py)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import datagen
from deepblast_amd.distributed import balanced_assignment, shard_bounds


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker_chunked(rank, world, port, outdir):
    """gather="e" with the backward sweep and the gather in pieces (uneven pieces: 5 pairs per rank in 3 chunks)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deepblast_amd import _engine, SmithWatermanDecoder
    from deepblast_amd.distributed import ShardedAligner
    from fake_engine import OracleEngine
    _engine._ENGINE = OracleEngine()
    B, N, M = 10, 20, 27
    theta, A = datagen.theta_A(47, B, N, M)
    lo, hi = shard_bounds(B, world, rank)
    res = {}
    for name, kw in (("chunked", dict(e_chunks=3)), ("async", dict(e_chunks=2, async_e=True)), ("one", dict(e_chunks=1))):
        al = ShardedAligner(SmithWatermanDecoder("softmax"), gather="e", **kw)
        out = al.align(torch.from_numpy(theta[lo:hi]), torch.from_numpy(A[lo:hi]))
        E = out["E"].wait() if kw.get("async_e") else out["E"]
        res[name + "_E"], res[name + "_Vt"], res[name + "_El"] = E.numpy(), out["Vt"].numpy(), out["E_local"].numpy()
        res[name + "_overlap"] = out["e_overlap"]
    np.savez(os.path.join(outdir, f"r{rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


def test_chunked_e_gather_world2(tmp_path):
    """E gathered in pieces under the backward sweep -- every rank ends up with the whole batch in batch
    order, identical to the one-collective gather; the result says which overlap was used."""
    import parity
    world = 2
    mp.spawn(_worker_chunked, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    theta, A = datagen.theta_A(47, 10, 20, 27)
    ref = parity.oracle_all(theta, A, None, None, 1, omp=False)
    for r in range(world):
        d = np.load(tmp_path / f"r{r}.npz")
        lo, hi = shard_bounds(10, world, r)
        for name in ("chunked", "async", "one"):
            assert np.array_equal(d[name + "_E"], ref["E"]), (r, name)
            assert np.array_equal(d[name + "_Vt"], ref["Vt"]), (r, name)
            assert np.array_equal(d[name + "_El"], ref["E"][lo:hi]), (r, name)
        assert str(d["chunked_overlap"]) == "chunked" and str(d["async_overlap"]) == "chunked" and str(d["one_overlap"]) == "none"


def _worker(rank, world, port, gather, outdir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deepblast_amd import _engine, NeedlemanWunschDecoder
    from deepblast_amd.distributed import ShardedAligner
    from fake_engine import OracleEngine
    _engine._ENGINE = OracleEngine()
    B, N, M = 6, 24, 31
    theta, A = datagen.theta_A(42, B, N, M)
    lo, hi = shard_bounds(B, world, rank)
    al = ShardedAligner(NeedlemanWunschDecoder("softmax"), gather=gather)
    out = al.align(torch.from_numpy(theta[lo:hi]), torch.from_numpy(A[lo:hi]))
    np.savez(os.path.join(outdir, f"r{rank}.npz"),
             Vt=out["Vt"].numpy() if out["Vt"] is not None else np.zeros(0),
             E=out["E"].numpy() if out["E"] is not None else np.zeros(0),
             Vt_local=out["Vt_local"].numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("gather", ["vt", "e"])
def test_sharded_align_world2(tmp_path, gather):
    import parity
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, gather, str(tmp_path)), nprocs=world, join=True)
    theta, A = datagen.theta_A(42, 6, 24, 31)
    ref = parity.oracle_all(theta, A, None, None, 0, omp=False)
    for r in range(world):
        d = np.load(tmp_path / f"r{r}.npz")
        assert np.array_equal(d["Vt"], ref["Vt"])          # every rank holds the full gathered result
        lo, hi = shard_bounds(6, world, r)
        assert np.array_equal(d["Vt_local"], ref["Vt"][lo:hi])
        if gather == "e":
            assert np.array_equal(d["E"], ref["E"])


def _worker_balanced(rank, world, port, outdir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deepblast_amd import _engine, NeedlemanWunschDecoder
    from deepblast_amd.distributed import BalancedPlan, ShardedAligner
    from fake_engine import OracleEngine
    _engine._ENGINE = OracleEngine()
    B, N, M = 7, 30, 26   # 7 pairs on 2 ranks: one rank pads with a dummy pair
    theta, A = datagen.theta_A(43, B, N, M)
    lens = datagen.lengths(44, B, 3, 26)
    plan = BalancedPlan(lens, world)
    mine = plan.indices(rank)
    al = ShardedAligner(NeedlemanWunschDecoder("softmax"), gather="e")
    out = al.align(torch.from_numpy(theta[mine]), torch.from_numpy(A[mine]), torch.from_numpy(lens[mine]), plan=plan)
    np.savez(os.path.join(outdir, f"r{rank}.npz"), Vt=out["Vt"].numpy(), E=out["E"].numpy(),
             Vt_local=out["Vt_local"].numpy(), mine=mine)
    dist.barrier()
    dist.destroy_process_group()


def test_balanced_plan_world2_restores_batch_order(tmp_path):
    """Variable-length batch: pairs dealt to ranks by work (LPT snake), uneven count padded with a dummy pair,
    gathered Vt and E come back in the ORIGINAL batch order."""
    import parity
    world = 2
    mp.spawn(_worker_balanced, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    theta, A = datagen.theta_A(43, 7, 30, 26)
    lens = datagen.lengths(44, 7, 3, 26)
    ref = parity.oracle_lens(theta, A, None, None, 0, lens)
    seen = []
    for r in range(world):
        d = np.load(tmp_path / f"r{r}.npz")
        assert d["Vt"].shape == (7,) and d["E"].shape == (7, 30, 26)
        assert np.array_equal(d["Vt"], ref["Vt"]) and np.array_equal(d["E"], ref["E"])
        assert np.array_equal(d["Vt_local"], ref["Vt"][d["mine"]])
        seen += d["mine"].tolist()
    assert sorted(seen) == list(range(7))


def _worker_paths(rank, world, port, outdir, rule="cpu"):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deepblast_amd import _engine, NeedlemanWunschDecoder
    from deepblast_amd.distributed import BalancedPlan, ShardedAligner
    from fake_engine import OracleEngine
    seen = []

    class Recording(OracleEngine):
        def traceback(self, grad, lens=None, rule="cpu"):
            seen.append(rule)
            return super().traceback(grad, lens, rule)
    _engine._ENGINE = Recording()
    B, N, M = 7, 30, 26
    theta, A = datagen.theta_A(45, B, N, M)
    theta = (theta * 6).astype(np.float32)   # peaked alignments: the walk follows a real path
    lens = datagen.lengths(46, B, 3, 26)
    plan = BalancedPlan(lens, world)
    mine = plan.indices(rank)
    al = ShardedAligner(NeedlemanWunschDecoder("softmax", traceback_rule=rule), gather="paths")
    out = al.align(torch.from_numpy(theta[mine]), torch.from_numpy(A[mine]), torch.from_numpy(lens[mine]), plan=plan)
    states, counts = out["paths"]
    np.savez(os.path.join(outdir, f"r{rank}.npz"), states=states.numpy(), counts=counts.numpy(), Vt=out["Vt"].numpy(), rules=np.array(seen))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("rule", ["cpu", "cuda"])
def test_gathered_paths_world2_match_per_pair_tracebacks(tmp_path, rule):
    """gather="paths": every rank ends up with the traceback of every pair, in batch order, equal to the reference's
    per-item decode + traceback (alignment.py:165-170) -- (N+M+2) int32 per pair over the wire instead of N x M floats.
    The walk is the DECODER's rule (traceback_rule="cuda": nw_cuda.py:273-317, stop as soon as one neighbour is off the
    matrix).  On alignment matrices the two rules give the same list whenever the CPU rule's walk stays on the matrix, so
    the test also checks which rule the engine was asked for."""
    import parity
    from deepblast_amd._dp import traceback
    world = 2
    mp.spawn(_worker_paths, args=(world, _free_port(), str(tmp_path), rule), nprocs=world, join=True)
    theta, A = datagen.theta_A(45, 7, 30, 26)
    theta = (theta * 6).astype(np.float32)
    lens = datagen.lengths(46, 7, 3, 26)
    ref = parity.oracle_lens(theta, A, None, None, 0, lens)
    for r in range(world):
        d = np.load(tmp_path / f"r{r}.npz")
        assert d["rules"].tolist() == [rule]
        assert d["states"].shape == (7, 30 + 26 + 2, 3) and np.array_equal(d["Vt"], ref["Vt"])
        for b in range(7):
            n, m = lens[b]
            want = traceback(ref["E"][b, :n, :m], rule=rule)
            assert d["counts"][b] == len(want)
            assert [tuple(int(v) for v in row) for row in d["states"][b, :len(want)]] == want


def test_pack_paths_round_trip_at_the_largest_shapes():
    from deepblast_amd.distributed import pack_paths, unpack_paths
    rng = np.random.default_rng(5)
    for N, M in ((131072, 2048), (7, 1), (1, 1), (513, 512)):
        cap = 40
        states = np.stack([rng.integers(0, N, (3, cap)), rng.integers(0, M, (3, cap)), rng.integers(0, 3, (3, cap))], axis=2).astype(np.int32)
        counts = np.array([cap, 5, -1], np.int32)
        s2, c2 = unpack_paths(pack_paths(torch.from_numpy(states), torch.from_numpy(counts), M))
        assert np.array_equal(c2.numpy(), counts)
        assert np.array_equal(s2.numpy()[0], states[0]) and np.array_equal(s2.numpy()[1, :5], states[1, :5])
        assert not s2.numpy()[2].any() and not s2.numpy()[1, 5:].any()   # rows past the count are zeroed


def test_balanced_plan_balances_work():
    from deepblast_amd.distributed import BalancedPlan
    lens = datagen.lengths(2, 2048, 64, 1024)
    plan = BalancedPlan(lens, 8)
    assert plan.per_rank == 256 and sorted(np.concatenate([plan.indices(r) for r in range(8)]).tolist()) == list(range(2048))
    assert plan.work_per_rank.max() / plan.work_per_rank.min() < 1.01
    g = torch.arange(8 * 256)   # a "gathered" tensor holding its own position
    back = plan.restore(g).numpy()
    for r in range(8):
        assert np.array_equal(back[plan.indices(r)], r * 256 + np.arange(256))


def test_shard_bounds_cover_batch():
    for B in (1, 7, 8, 256, 2048):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(B, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert [shard_bounds(2048, 8, r) for r in (0, 7)] == [(0, 256), (1792, 2048)]


def test_balanced_assignment_is_a_balanced_permutation():
    lens = datagen.lengths(3, 256, 64, 1024)
    work = lens[:, 0].astype(np.int64) * lens[:, 1]
    order, counts, inverse = balanced_assignment(work, 8)
    assert sorted(order.tolist()) == list(range(256))
    assert np.array_equal(order[inverse], np.arange(256))
    assert counts.tolist() == [32] * 8
    per_rank = [work[order[r * 32:(r + 1) * 32]].sum() for r in range(8)]
    assert max(per_rank) / min(per_rank) < 1.05
