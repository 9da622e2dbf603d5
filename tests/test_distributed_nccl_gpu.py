"""GPU, needs >= 2 devices (skipped on the 1-GPU box): the N>1 path on RCCL -- two ranks, one per GPU, each
aligning its shard with the HIP engine; gathered Vt / E against the CPU oracle, contiguous and balanced shards."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import datagen

pytestmark = pytest.mark.gpu


def _ndev():
    try:
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


needs2 = pytest.mark.skipif(_ndev() < 2, reason="needs >= 2 ROCm devices")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, mode, outdir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from deepblast_amd import NeedlemanWunschDecoder
    from deepblast_amd.distributed import BalancedPlan, ShardedAligner, shard_bounds
    B, N, M = 10, 150, 130
    theta, A = datagen.theta_A(45, B, N, M)
    al = ShardedAligner(NeedlemanWunschDecoder("softmax"), gather="e", async_e=(mode == "balanced"),
                        e_chunks=3 if mode == "chunked" else 1)
    if mode in ("contiguous", "chunked"):
        lo, hi = shard_bounds(B, world, rank)
        out = al.align(torch.from_numpy(theta[lo:hi]).to(dev), torch.from_numpy(A[lo:hi]).to(dev))
        E = out["E"]
    else:
        lens = datagen.lengths(46, B, 5, 130)
        plan = BalancedPlan(lens, world)
        mine = plan.indices(rank)
        out = al.align(torch.from_numpy(theta[mine]).to(dev), torch.from_numpy(A[mine]).to(dev),
                       torch.from_numpy(lens[mine]).to(dev), plan=plan)
        E = out["E"].wait()   # asynchronous gather handle
    torch.cuda.synchronize()
    np.savez(os.path.join(outdir, f"r{rank}.npz"), Vt=out["Vt"].cpu().numpy(), E=E.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


@needs2
@pytest.mark.parametrize("mode", ["contiguous", "balanced", "chunked"])
def test_sharded_align_two_ranks_rccl(tmp_path, mode):
    import parity
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    theta, A = datagen.theta_A(45, 10, 150, 130)
    if mode in ("contiguous", "chunked"):
        ref = parity.oracle_all(theta, A, None, None, 0, omp=False)
    else:
        ref = parity.oracle_lens(theta, A, None, None, 0, datagen.lengths(46, 10, 5, 130))
    for r in range(world):
        d = np.load(tmp_path / f"r{r}.npz")
        assert parity.rel_err(d["Vt"], ref["Vt"]) <= parity.TOL
        assert parity.abs_err(d["E"], ref["E"]) <= parity.TOL


@pytest.mark.parametrize("mode", ["contiguous", "balanced", "chunked"])
def test_sharded_align_one_rank_rccl(tmp_path, mode):
    """What a 1-GPU box CAN run of the RCCL path: a process group of one rank on the nccl backend.  Every collective the
    sharded path issues -- all_gather_into_tensor of Vt under the backward sweep, the E gather (one collective, the
    asynchronous handle, the pieces under the sweep), the barrier -- goes through RCCL with the real argument types,
    streams and async handles; what it cannot show is a second rank."""
    import parity
    mp.spawn(_worker, args=(1, _free_port(), mode, str(tmp_path)), nprocs=1, join=True)
    theta, A = datagen.theta_A(45, 10, 150, 130)
    if mode in ("contiguous", "chunked"):
        ref = parity.oracle_all(theta, A, None, None, 0, omp=False)
    else:
        ref = parity.oracle_lens(theta, A, None, None, 0, datagen.lengths(46, 10, 5, 130))
    d = np.load(tmp_path / "r0.npz")
    assert parity.rel_err(d["Vt"], ref["Vt"]) <= parity.TOL
    assert parity.abs_err(d["E"], ref["E"]) <= parity.TOL


def _worker_paths(rank, world, port, backend, outdir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    idx = rank if backend == "nccl" else 0   # gloo: both ranks share GPU 0 (what a 1-GPU box can check)
    torch.cuda.set_device(idx)
    dev = torch.device("cuda", idx)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from deepblast_amd import NeedlemanWunschDecoder
    from deepblast_amd.distributed import BalancedPlan, ShardedAligner
    B, N, M = 9, 150, 130
    theta, A = datagen.theta_A(47, B, N, M)
    theta = (theta * 6).astype(np.float32)
    lens = datagen.lengths(48, B, 5, 130)
    plan = BalancedPlan(lens, world)
    mine = plan.indices(rank)
    al = ShardedAligner(NeedlemanWunschDecoder("softmax"), gather="paths")
    out = al.align(torch.from_numpy(theta[mine]).to(dev), torch.from_numpy(A[mine]).to(dev),
                   torch.from_numpy(lens[mine]).to(dev), plan=plan)
    states, counts = out["paths"]
    torch.cuda.synchronize()
    np.savez(os.path.join(outdir, f"r{rank}.npz"), states=states.cpu().numpy(), counts=counts.cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


def _check_paths(tmp_path, world):
    import parity
    from deepblast_amd._dp import traceback
    theta, A = datagen.theta_A(47, 9, 150, 130)
    theta = (theta * 6).astype(np.float32)
    lens = datagen.lengths(48, 9, 5, 130)
    ref = parity.oracle_lens(theta, A, None, None, 0, lens)
    for r in range(world):
        d = np.load(tmp_path / f"r{r}.npz")
        for b in range(9):
            n, m = lens[b]
            want = traceback(ref["E"][b, :n, :m])
            assert d["counts"][b] == len(want), (r, b)
            assert [tuple(int(v) for v in row) for row in d["states"][b, :len(want)]] == want, (r, b)


def test_gathered_paths_two_ranks_share_one_gpu(tmp_path):
    """gather="paths" with the HIP engine and the device traceback kernel, two ranks on GPU 0 over gloo: every rank holds
    every pair's traceback, in batch order, equal to the per-pair host walk over the oracle's E."""
    mp.spawn(_worker_paths, args=(2, _free_port(), "gloo", str(tmp_path)), nprocs=2, join=True)
    _check_paths(tmp_path, 2)


def test_gathered_paths_one_rank_rccl(tmp_path):
    """The walks gathered through RCCL by a group of one rank (see test_sharded_align_one_rank_rccl)."""
    mp.spawn(_worker_paths, args=(1, _free_port(), "nccl", str(tmp_path)), nprocs=1, join=True)
    _check_paths(tmp_path, 1)


@needs2
def test_gathered_paths_two_ranks_rccl(tmp_path):
    mp.spawn(_worker_paths, args=(2, _free_port(), "nccl", str(tmp_path)), nprocs=2, join=True)
    _check_paths(tmp_path, 2)


@pytest.mark.parametrize("B,N,M,pieces", [(37, 130, 200, ((0, 5), (5, 6), (6, 30))),
                                          (12, 512, 512, ((0, 3), (3, 4), (4, 11))),       # N > 256: a pair's share of the buffer's TAIL (bridge rows) is not part of its record
                                          (16, 1024, 1024, ((0, 4), (4, 8), (8, 12), (12, 16))),   # the whole-batch launch spreads pairs over workgroups, the pieces must not
                                          (5, 1100, 300, ((0, 2), (2, 5)))])
def test_backward_sweep_in_pieces_is_bit_identical(B, N, M, pieces):
    """The backward sweep of a batch launched in pieces (HipEngine.backward(pair_range=, out=) -> sdp_backward_range_f32:
    what the chunked E gather does) writes exactly what one launch writes, packed and exact state, leaves the other rows
    alone -- and leaves the STATE alone: a second whole-batch sweep after the pieces still gives the same E (round 3
    located the pieces' records by sdp_state_bytes(2) - sdp_state_bytes(1), 5440 bytes per pair off at 512 x 512, and a
    piece that chose the several-workgroups schedule reset "its" bridge rows inside the next pairs' records)."""
    from deepblast_amd._engine import get_engine
    eng = get_engine()
    theta, A = datagen.theta_A(49, B, N, M)
    t, a = torch.from_numpy(theta).cuda(), torch.from_numpy(A).cuda()
    et = torch.from_numpy((0.5 + datagen.uniform(50, (B,))).astype(np.float32)).cuda()
    covered = max(hi for _, hi in pieces)
    for exact in (False, True):
        for variant in (0, 1):
            Vt, Q = eng.forward(t, a, variant, exact_state=exact)
            whole = eng.backward(et, Q, (B, N, M), variant, exact_state=exact)
            out = torch.full((B, N, M), -7.0, device="cuda")
            for lo, hi in pieces:
                eng.backward(et, Q, (B, N, M), variant, exact_state=exact, pair_range=(lo, hi), out=out)
            assert torch.equal(out[:covered], whole[:covered]) and bool((out[covered:] == -7.0).all())
            assert torch.equal(eng.backward(et, Q, (B, N, M), variant, exact_state=exact), whole)
    assert eng.state_pair_bytes(N, M) * B <= eng.lib.sdp_state_bytes(B, N, M)
    with pytest.raises(ValueError):
        eng.backward(et, Q, (B, N, M), 0, lens=torch.ones(B, 2, dtype=torch.int32, device="cuda"), pair_range=(0, 2), out=out)
    with pytest.raises(ValueError):
        eng.backward(et, Q, (B, N, M), 0, pair_range=(B - 1, B + 1), out=out)


def test_broadcast_cotangent_equals_expanded():
    """SDP_ET_BROADCAST: a one-element Et (what Vt.sum().backward() hands over, as a stride-0 view) gives the same E as the
    expanded (B,) tensor, without the expand-and-copy kernel in between."""
    from deepblast_amd._engine import get_engine
    eng = get_engine()
    B, N, M = 9, 70, 130
    theta, A = datagen.theta_A(51, B, N, M)
    t, a = torch.from_numpy(theta).cuda(), torch.from_numpy(A).cuda()
    _, Q = eng.forward(t, a, 0)
    one = torch.full((1,), 0.75, device="cuda")
    want = eng.backward(one.expand(B).contiguous(), Q, (B, N, M), 0)
    assert torch.equal(eng.backward(one.expand(B), Q, (B, N, M), 0), want)
    assert torch.equal(eng.backward(one[0], Q, (B, N, M), 0), want)
    out = torch.zeros_like(want)
    eng.backward(one.expand(B), Q, (B, N, M), 0, pair_range=(0, 4), out=out)
    eng.backward(one.expand(B), Q, (B, N, M), 0, pair_range=(4, B), out=out)
    assert torch.equal(out, want)


def test_bench_refuses_more_gpus_than_present():
    """`python bench.py --gpus N` creates its own ranks and must fail loudly -- not time fewer GPUs -- when the node
    has fewer than N devices."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    n = _ndev() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "device(s) visible" in (r.stderr + r.stdout)
    assert '"n_gpus"' not in r.stdout


def test_bench_two_ranks_flow_on_one_gpu():
    """The N > 1 flow of bench.py (rank setup, sharded step, Vt gather under the backward sweep, max-over-ranks timing,
    the secondary E-gather figure, one JSON line from rank 0) exercised with two ranks that share GPU 0 and talk over
    gloo -- what a 1-GPU box can check of the path the driver runs on 8 GPUs over RCCL."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(BENCH_SHARE_GPU="1", BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--B", "24", "--N", "200", "--M", "180"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["config"]["global_batch"] == 48 and d["config"]["gather"] == "vt"
    assert d["value"] > 0 and d["with_e_gather"]["value"] > 0 and "cpu_baseline" not in d
    assert d["with_paths_gather"]["value"] > 0 and d["with_paths_gather"]["bytes_into_each_gpu"] < d["with_e_gather"]["bytes_into_each_gpu"] / 50
    assert d["scaling"] == "weak" and "test mode" in d["config"]["backend"]


def test_bench_multi_gpu_flow_one_rank_rccl():
    """bench.py's N > 1 flow with a world of one rank on the nccl backend (BENCH_FORCE_DIST=1): the process group on RCCL, the
    Vt gather under the backward sweep, the barrier fences, the all-reduce of the timing, the secondary E-gather and
    path-gather measurements -- every collective of the flow the driver runs on 8 GPUs, through RCCL, on one GPU."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("BENCH_SHARE_GPU", "BENCH_BACKEND")}
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), BENCH_FORCE_DIST="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--B", "24", "--N", "200",
           "--M", "180", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["config"]["backend"] == "rccl" and d["config"]["gather"] == "vt"
    assert d["value"] > 0 and d["with_e_gather"]["value"] > 0 and d["with_paths_gather"]["value"] > 0
