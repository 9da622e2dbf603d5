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
