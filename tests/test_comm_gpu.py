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
