"""GPU (one device is enough): the N = 8 path of BASELINE.json configs[4] -- B = 2048 pairs of 512 x 512, 256 per rank,
seed 3 -- with EIGHT RANKS SHARING GPU 0 and talking over gloo.  What a 1-GPU box can check of the job the driver runs on
8 GPUs over RCCL: the HIP engine on every rank, the contiguous sharding, the Vt / walks / E gathers (E in one collective
and in four pieces under the backward sweep), batch order, and every one of the 2048 results against the CPU oracle.
The sweeps being sharded are the reference's serial `for b in range(B)` (deepblast/nw.py:110-115); its only
parallelism is Lightning DDP (scripts/deepblast-train:66-76).
"""
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import datagen

pytestmark = pytest.mark.gpu

WORLD, B_RANK, N, M, SEED = 8, 256, 512, 512, 3   # BASELINE.json configs[4]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(fn, args, nprocs, timeout):
    """mp.spawn with a deadline: a rank that hangs in a collective must fail the test, not the test run."""
    ctx = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
    t0 = time.time()
    while not ctx.join(timeout=5):
        if time.time() - t0 > timeout:
            for p in ctx.processes:
                if p.is_alive():
                    p.kill()
            raise AssertionError(f"ranks did not finish within {timeout} s")


def _worker(rank, world, port, outdir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deepblast_amd import NeedlemanWunschDecoder
    from deepblast_amd.distributed import ShardedAligner, shard_bounds
    Bg = world * B_RANK
    lo, hi = shard_bounds(Bg, world, rank)
    theta, A = datagen.theta_A(SEED, Bg, N, M, rows=(lo, hi))
    t, a = torch.from_numpy(theta).to(dev), torch.from_numpy(A).to(dev)
    dec = NeedlemanWunschDecoder("softmax")
    ref_Vt = torch.from_numpy(np.load(os.path.join(outdir, "ref_Vt.npy"))).to(dev)
    ref_E = np.load(os.path.join(outdir, "ref_E.npy"), mmap_mode="r")
    ref_states = np.load(os.path.join(outdir, "ref_states.npy"))
    ref_counts = np.load(os.path.join(outdir, "ref_counts.npy"))
    res = {}

    def vt_err(Vt):
        return float(((Vt - ref_Vt).abs() / ref_Vt.abs().clamp(min=1.0)).max())

    def e_err(E):
        worst = 0.0
        for c in range(0, Bg, 128):
            worst = max(worst, float((E[c:c + 128] - torch.from_numpy(np.ascontiguousarray(ref_E[c:c + 128])).to(dev)).abs().max()))
        return worst

    # terminal scores only (the headline job), in the metric's own idiom
    out = ShardedAligner(dec, gather="vt", idiom="sum_backward").align(t, a)
    res["vt_Vt"] = vt_err(out["Vt"])
    res["vt_Elocal"] = float((out["E_local"] - torch.from_numpy(np.ascontiguousarray(ref_E[lo:hi])).to(dev)).abs().max())
    # walks gathered instead of matrices
    out = ShardedAligner(dec, gather="paths").align(t, a)
    states, counts = out["paths"]
    res["paths_Vt"] = vt_err(out["Vt"])
    counts, states = counts.cpu().numpy(), states.cpu().numpy()
    # matrices gathered: in four pieces under the backward sweep, and in one collective after it
    out4 = ShardedAligner(dec, gather="e", e_chunks=4).align(t, a)
    res["e4_overlap"] = out4["e_overlap"]
    res["e4_Vt"] = vt_err(out4["Vt"])
    res["e4_E"] = e_err(out4["E"])
    E4 = out4["E"]
    del out4
    out1 = ShardedAligner(dec, gather="e", e_chunks=1).align(t, a)
    res["e1_overlap"] = out1["e_overlap"]
    res["e1_equals_e4"] = bool(torch.equal(out1["E"], E4)) and bool(torch.equal(out1["Vt"], out["Vt"]))
    # the gathered walks are the walks over the gathered (and oracle-checked) matrices, pair by pair in batch order ...
    from deepblast_amd._engine import get_engine
    st_e, cn_e = get_engine().traceback(E4)
    st_e, cn_e = st_e.cpu().numpy(), cn_e.cpu().numpy()
    same = np.array_equal(counts, cn_e)
    for b in range(Bg):
        same = same and np.array_equal(states[b, :counts[b]], st_e[b, :cn_e[b]])
    res["paths_equal_walks_over_gathered_E"] = bool(same)
    # ... and, except where two neighbours tie to within the engine's rounding (soft random scores: E is diffuse and a
    # greedy arg-max over it is decided by differences far below 1e-4), the host walks over the ORACLE's matrices
    agree = sum(int(counts[b] == ref_counts[b] and np.array_equal(states[b, :counts[b]], ref_states[b, :ref_counts[b]])) for b in range(Bg))
    res["paths_agree_with_oracle_walks"] = agree / Bg
    torch.cuda.synchronize()
    with open(os.path.join(outdir, f"r{rank}.json"), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


def test_configs4_eight_ranks_share_one_gpu(tmp_path):
    import parity
    from deepblast_amd._dp import traceback
    from deepblast_amd.distributed import shard_bounds
    Bg = WORLD * B_RANK
    # the oracle's answer for all 2048 pairs (OpenMP over the host's cores), written where the ranks can map it
    ref_E = np.lib.format.open_memmap(tmp_path / "ref_E.npy", mode="w+", dtype=np.float32, shape=(Bg, N, M))
    ref_Vt = np.zeros(Bg, np.float32)
    cap = N + M + 2
    ref_states = np.zeros((Bg, cap, 3), np.int32)
    ref_counts = np.zeros(Bg, np.int32)
    for r in range(WORLD):
        lo, hi = shard_bounds(Bg, WORLD, r)
        theta, A = datagen.theta_A(SEED, Bg, N, M, rows=(lo, hi))
        ref = parity.oracle_chunked(theta, A, None, None, 0, chunk=64)
        ref_E[lo:hi] = ref["E"]
        ref_Vt[lo:hi] = ref["Vt"]
        for b in range(lo, hi):
            path = traceback(ref["E"][b - lo])
            ref_counts[b] = len(path)
            ref_states[b, :len(path)] = np.asarray(path, np.int32)
    ref_E.flush()
    del ref_E
    np.save(tmp_path / "ref_Vt.npy", ref_Vt)
    np.save(tmp_path / "ref_states.npy", ref_states)
    np.save(tmp_path / "ref_counts.npy", ref_counts)
    _spawn(_worker, (WORLD, _free_port(), str(tmp_path)), WORLD, timeout=1500)
    for r in range(WORLD):
        d = json.load(open(tmp_path / f"r{r}.json"))
        assert d["vt_Vt"] <= parity.TOL and d["paths_Vt"] <= parity.TOL and d["e4_Vt"] <= parity.TOL, (r, d)
        assert d["vt_Elocal"] <= parity.TOL and d["e4_E"] <= parity.TOL, (r, d)
        assert d["paths_equal_walks_over_gathered_E"], (r, d)
        assert d["paths_agree_with_oracle_walks"] >= 0.95, (r, d)
        assert d["e4_overlap"] == "chunked" and d["e1_overlap"] == "none" and d["e1_equals_e4"], (r, d)


def _run_bench(nproc, extra_args, extra_env, timeout):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(BENCH_SHARE_GPU="1", BENCH_BACKEND="gloo")
    env.update(extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(nproc)] + extra_args
    return subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=timeout)


def test_bench_eight_ranks_flow_on_one_gpu():
    """`bench.py --gpus 8` exactly as the driver launches it (torch.distributed.run, 8 ranks, default shape = 256 pairs of
    512 x 512 per rank = BASELINE configs[4]) -- except that the ranks share GPU 0 and gather over gloo: exactly one JSON
    line, n_gpus 8, global batch 2048, the secondary gather figures present."""
    # (--e-chunks 4: the E gather in pieces under the backward sweep is exercised here; the default is one collective since round 5)
    r = _run_bench(8, ["--steps", "2", "--warmup", "1", "--e-chunks", "4"], {}, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 2 and d["config"]["global_batch"] == 2048 and d["config"]["gather"] == "vt"
    assert "configs[4]" in d["config"]["workload"] and "Vt.sum().backward()" in d["config"]["step"]
    assert d["value"] > 0 and d["direct_cotangent"]["value"] > 0 and "cpu_baseline" not in d
    assert d["with_e_gather"]["value"] > 0 and d["with_e_gather"]["e_chunks"] == 4 and d["with_e_gather"]["one_collective_ms_per_step"] > 0
    assert d["with_paths_gather"]["value"] > 0
    assert d["scaling"] == "weak" and "test mode" in d["config"]["backend"]


def test_bench_primary_line_survives_a_hung_secondary():
    """A rank that never reaches the collective of a SECONDARY measurement (here: on purpose) must not cost the primary
    figure: the watchdog lets rank 0 print the one JSON line without the secondary fields, and every rank exits."""
    r = _run_bench(2, ["--steps", "2", "--warmup", "1", "--B", "24", "--N", "200", "--M", "180"],
                   {"BENCH_TEST_HANG_RANK": "1", "BENCH_WATCHDOG_S": "20"}, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["global_batch"] == 48
    assert "with_e_gather" not in d and "direct_cotangent" not in d
    assert "did not finish in time" in r.stderr
