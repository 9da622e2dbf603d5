"""GPU: error paths and re-entrancy of the C ABI -- a strip hand-off that times out is REPORTED (SDP_E_HANDOFF), and
calls from two threads on two streams may overlap."""
import ctypes
import os
import threading

import numpy as np
import pytest
import torch

import datagen
import parity

pytestmark = pytest.mark.gpu


def _exp_lib():
    from deepblast_amd import _lib, build
    try:
        path = build.build_experiments()   # rebuilds only if the sources are newer than the library (hipcc, ~40 s)
    except Exception as e:   # no compiler on this box: use what travelled with the snapshot
        path = build.EXP_OUT
        if not os.path.exists(path):
            pytest.skip(f"deepblast_amd/libsdp_hip_exp.so not built and cannot be built here ({e})")
    return _lib.load_path(path)


def test_handoff_timeout_is_reported_not_silent():
    """Experiments build, debug bit 3: strips never publish their progress, so the strip below gives up waiting.
    The library must say so -- sdp_device_status / the next call return SDP_E_HANDOFF once -- instead of returning
    0 with wrong numbers."""
    lib = _exp_lib()
    B, N, M = 2, 200, 100   # 4 strips per pair
    theta, A = datagen.theta_A(5, B, N, M)
    dev = torch.device("cuda", 0)
    t, a = torch.from_numpy(theta).to(dev), torch.from_numpy(A).to(dev)
    state = torch.empty(lib.sdp_state_bytes(B, N, M) // 4, dtype=torch.float32, device=dev)
    vt = torch.empty(B, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(0).cuda_stream
    info = (ctypes.c_int32 * 4)()

    def fwd():
        return lib.sdp_forward_f32(t.data_ptr(), a.data_ptr(), state.data_ptr(), vt.data_ptr(), B, N, M, None, 0, 0, stream)

    assert fwd() == 0
    torch.cuda.synchronize()
    assert lib.sdp_device_status(0, info) == 0 and info[0] == 0
    good = vt.cpu().numpy().copy()
    ref = parity.oracle_all(theta, A, None, None, 0)
    assert parity.rel_err(good, ref["Vt"]) <= parity.TOL

    assert lib.sdp_set_debug(8) == 0
    try:
        assert fwd() == 0            # the launch itself is fine; the kernel reports while it runs
        torch.cuda.synchronize()
    finally:
        lib.sdp_set_debug(0)
    rc = fwd()                       # the NEXT call on the device reports it, once
    assert rc == -7, rc
    msg = lib.sdp_last_error_string().decode()
    assert "hand-off timed out" in msg and "pair" in msg
    assert lib.sdp_device_status(0, info) == 0   # already reported
    assert info[0] >= 1 and 0 <= info[1] < B and 1 <= info[2] < 4 and (info[3] >> 24) == 0
    assert fwd() == 0                # and the library keeps working
    torch.cuda.synchronize()
    assert np.array_equal(vt.cpu().numpy(), good)


def test_two_threads_two_streams_overlap():
    """The ABI keeps no per-call global state: two host threads drive all four sweeps on their own streams at the
    same time, each with its own inputs; results equal the single-threaded ones bit for bit."""
    from deepblast_amd._engine import get_engine
    eng = get_engine()
    dev = torch.device("cuda", 0)
    jobs = []
    for k, (B, N, M, variant) in enumerate([(40, 300, 200, 0), (24, 130, 520, 1)]):
        theta, A = datagen.theta_A(900 + k, B, N, M)
        Z = datagen.normal(910 + k, (B, N, M))
        jobs.append((theta, A, Z, variant, parity.engine_all(theta, A, None, Z, variant)))
    results = [None, None]
    errors = []

    def run(i):
        try:
            theta, A, Z, variant, _ = jobs[i]
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                out = None
                for _ in range(8):
                    out = parity.engine_all(theta, A, None, Z, variant)
            s.synchronize()
            results[i] = out
        except Exception as e:  # surfaced below
            errors.append(e)

    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors
    for i in range(2):
        for k, v in jobs[i][4].items():
            assert np.array_equal(results[i][k], v), (i, k)
    assert eng.check_device()[0] == 0
