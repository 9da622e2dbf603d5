"""GPU: the RCCL wrappers of the C ABI (sdp_comm_*): the one collective of the multi-GPU path -- gathering results --
for callers that do not use torch.distributed.  One rank runs on any GPU box; the two-rank case needs two devices."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from deepblast_amd import _lib
    return _lib.load()


def test_single_rank_all_gather_is_a_copy():
    lib = _lib()
    ident = (ctypes.c_char * 128)()
    assert lib.sdp_comm_unique_id(ident) == 0, lib.sdp_comm_last_error_string()
    comm = ctypes.c_void_p()
    assert lib.sdp_comm_init(ctypes.byref(comm), ident, 0, 1, 0) == 0, lib.sdp_comm_last_error_string()
    send = torch.arange(1000, dtype=torch.float32, device="cuda:0") * 0.5
    recv = torch.zeros(1000, dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream(0).cuda_stream
    assert lib.sdp_comm_all_gather_f32(comm, send.data_ptr(), recv.data_ptr(), 1000, stream) == 0
    torch.cuda.synchronize()
    assert torch.equal(send, recv)
    assert lib.sdp_comm_destroy(comm) == 0
    assert lib.sdp_comm_init(None, ident, 0, 1, 0) == -1 and b"null" in lib.sdp_comm_last_error_string()
    assert lib.sdp_comm_init(ctypes.byref(comm), ident, 3, 2, 0) == -2


def _rank(rank, world, idfile, outdir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import time
    import datagen
    import parity
    from deepblast_amd import _lib
    from deepblast_amd._engine import get_engine
    lib = _lib.load()
    torch.cuda.set_device(rank)
    ident = (ctypes.c_char * 128)()
    if rank == 0:
        assert lib.sdp_comm_unique_id(ident) == 0
        with open(idfile + ".tmp", "wb") as f:
            f.write(bytes(ident))
        os.replace(idfile + ".tmp", idfile)
    else:
        for _ in range(600):
            if os.path.exists(idfile):
                break
            time.sleep(0.05)
        ident = (ctypes.c_char * 128).from_buffer_copy(open(idfile, "rb").read())
    comm = ctypes.c_void_p()
    assert lib.sdp_comm_init(ctypes.byref(comm), ident, rank, world, rank) == 0, lib.sdp_comm_last_error_string()
    # this rank's shard through the engine (no torch.distributed anywhere), then the gather of Vt
    B, N, M = 6, 90, 70
    theta, A = datagen.theta_A(77, B, N, M)
    lo, hi = rank * 3, rank * 3 + 3
    dev = torch.device("cuda", rank)
    Vt, _ = get_engine().forward(torch.from_numpy(theta[lo:hi]).to(dev), torch.from_numpy(A[lo:hi]).to(dev), 0)
    out = torch.empty(B, dtype=torch.float32, device=dev)
    assert lib.sdp_comm_all_gather_f32(comm, Vt.data_ptr(), out.data_ptr(), 3, torch.cuda.current_stream(rank).cuda_stream) == 0
    torch.cuda.synchronize()
    np.save(os.path.join(outdir, f"vt{rank}.npy"), out.cpu().numpy())
    assert lib.sdp_comm_destroy(comm) == 0


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs >= 2 ROCm devices")
def test_two_ranks_gather_scores(tmp_path):
    import torch.multiprocessing as mp
    import datagen
    import parity
    mp.spawn(_rank, args=(2, str(tmp_path / "id.bin"), str(tmp_path)), nprocs=2, join=True)
    theta, A = datagen.theta_A(77, 6, 90, 70)
    ref = parity.oracle_all(theta, A, None, None, 0, omp=False)
    for r in range(2):
        assert parity.rel_err(np.load(tmp_path / f"vt{r}.npy"), ref["Vt"]) <= parity.TOL


def _rank_shared(rank, world, idfile, outdir):
    """Two ranks on ONE device: the id hand-off between processes is real, the communicator is not expected to come up."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import json
    import time
    from deepblast_amd import _lib
    lib = _lib.load()
    torch.cuda.set_device(0)
    ident = (ctypes.c_char * 128)()
    if rank == 0:
        assert lib.sdp_comm_unique_id(ident) == 0
        with open(idfile + ".tmp", "wb") as f:
            f.write(bytes(ident))
        os.replace(idfile + ".tmp", idfile)
    else:
        for _ in range(600):
            if os.path.exists(idfile):
                break
            time.sleep(0.05)
        ident = (ctypes.c_char * 128).from_buffer_copy(open(idfile, "rb").read())
    comm = ctypes.c_void_p()
    rc = lib.sdp_comm_init(ctypes.byref(comm), ident, rank, world, 0)
    res = {"rc": rc, "error": lib.sdp_comm_last_error_string().decode("utf-8", "replace") if rc else ""}
    if rc == 0:   # RCCL accepted two ranks on one device: then the gather must work too
        send = torch.full((5,), float(rank + 1), device="cuda:0")
        recv = torch.zeros(10, device="cuda:0")
        assert lib.sdp_comm_all_gather_f32(comm, send.data_ptr(), recv.data_ptr(), 5, torch.cuda.current_stream(0).cuda_stream) == 0
        torch.cuda.synchronize()
        res["gathered"] = recv.cpu().tolist()
        lib.sdp_comm_destroy(comm)
    with open(os.path.join(outdir, f"shared{rank}.json"), "w") as f:
        json.dump(res, f)


def test_two_ranks_sharing_one_device():
    """What a 1-GPU box can exercise of sdp_comm_* with more than one rank: the unique id made by rank 0 and handed to
    rank 1 through a file, both calling sdp_comm_init.  RCCL (like NCCL) refuses a communicator with two ranks on the same
    device ("Duplicate GPU detected"): the wrapper must then return SDP_E_COMM with that message on every rank -- not hang,
    not crash -- which is also why the shared-GPU tests of the sharded path (tests/test_multirank_one_gpu.py) gather over gloo.
    Should a future RCCL accept it, the gather itself is checked instead."""
    import json
    import tempfile
    import time
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as tmp:
        ctx = mp.spawn(_rank_shared, args=(2, os.path.join(tmp, "id.bin"), tmp), nprocs=2, join=False)
        t0 = time.time()
        while not ctx.join(timeout=5):
            if time.time() - t0 > 180:
                for p in ctx.processes:
                    if p.is_alive():
                        p.kill()
                raise AssertionError("sdp_comm_init with two ranks on one device hung")
        res = [json.load(open(os.path.join(tmp, f"shared{r}.json"))) for r in range(2)]
    print("two ranks on one device:", res)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "comm_shared_device.txt"), "w") as f:
            f.write(json.dumps(res) + "\n")
    if all(r["rc"] == 0 for r in res):
        for r in res:
            assert r["gathered"] == [1.0] * 5 + [2.0] * 5
    else:
        assert all(r["rc"] == -8 and r["error"] for r in res), res   # SDP_E_COMM with RCCL's own message
