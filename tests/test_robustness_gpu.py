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


def test_walks_on_a_second_stream_beside_the_next_batch_s_sweeps():
    """Inference as a pipeline (bench.py --mode align+traceback, `pipelined_walk`): the batched walk of batch k runs on a second
    stream while the first stream already sweeps batch k + 1.  Every batch's walks equal the ones of the serial order."""
    from deepblast_amd._engine import get_engine
    eng = get_engine()
    B, N, M = 96, 200, 260
    batches = []
    for k in range(3):
        theta, A = datagen.theta_A(1300 + k, B, N, M)
        batches.append((torch.from_numpy(theta).cuda(), torch.from_numpy(A).cuda()))
    ones = torch.ones(B, device="cuda")

    def sweeps(t, a):
        Vt, Q = eng.forward(t, a, 0)
        return eng.backward(ones, Q, tuple(t.shape), 0)

    serial = []
    for t, a in batches:
        st, cn = eng.traceback(sweeps(t, a))
        serial.append((st.cpu().numpy(), cn.cpu().numpy()))
    torch.cuda.synchronize()
    ws = torch.cuda.Stream()
    piped = []
    for rnd in range(4):                       # several rounds: the allocator gets to reuse E's memory
        for t, a in batches:
            E = sweeps(t, a)
            ws.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(ws):
                st, cn = eng.traceback(E)
            E.record_stream(ws)
            del E
            piped.append((st, cn))
    torch.cuda.synchronize()
    for i, (st, cn) in enumerate(piped):
        want_st, want_cn = serial[i % len(batches)]
        got_cn = cn.cpu().numpy()
        assert np.array_equal(got_cn, want_cn), i
        got_st = st.cpu().numpy()
        for b in range(B):
            assert np.array_equal(got_st[b, :got_cn[b]], want_st[b, :want_cn[b]]), (i, b)
    assert eng.check_device()[0] == 0


def test_pairs_over_several_workgroups_repeat_and_overlap():
    """Parts of a pair hand their boundary over through global memory with no fence and no flag (8-byte granules that
    are either the memset pattern or written, sdp_kernels.hip "bridge"), in an order that must never leave a consumer on
    a CU without its producer.  A protocol like that can be right by luck: run a batch with per-pair lengths that takes
    it (the library's own policy) forty times, the second half of them while another stream streams through memory and a
    second batch with its own state runs its sweeps on a third stream; every run must return the same bits, equal the
    one-workgroup schedule, and no hand-off may have timed out."""
    from deepblast_amd._engine import get_engine
    eng = get_engine()
    B, N, M = 200, 960, 640
    # (the library's own policy: with per-pair lengths the FORWARD sweep takes parts here; since round 5 the backward sweep
    #  takes them for a few long EQUAL pairs only -- the second problem below is such a batch, so both bridges stay under test)
    assert eng.lib.sdp_plan_parts(0, B, N, M, 1, 0, 256) == 4 and eng.lib.sdp_plan_parts(1, 16, 1024, M, 0, 0, 256) == 4
    theta, A = datagen.theta_A(93001, B, N, M)
    lens = datagen.lengths(93002, B, 1, N)
    lens[:, 1] = np.minimum(lens[:, 1] * M // N + 1, M)
    lens[0] = (N, M)
    t, a = torch.from_numpy(theta).cuda(), torch.from_numpy(A).cuda()
    ln = torch.from_numpy(lens).cuda()
    et = torch.from_numpy((0.5 + datagen.uniform(93003, (B,))).astype(np.float32)).cuda()
    # a second problem that takes parts too, for the third stream: 16 equal pairs of 1024 x 640 (backward sweep in parts)
    th2, A2 = datagen.theta_A(93004, 16, 1024, M)
    t2, a2, ln2 = torch.from_numpy(th2).cuda() * 0.5, torch.from_numpy(A2).cuda(), None
    s_noise, s_other = torch.cuda.Stream(), torch.cuda.Stream()
    big = torch.empty(64 * 1024 * 1024, device="cuda")

    def sweep(tt, aa, ll, ee):
        Vt, Q = eng.forward(tt, aa, 0, ll)
        return Vt, eng.backward(ee, Q, tuple(tt.shape), 0, ll)

    first = other_first = None
    for it in range(40):
        if it >= 20:
            s_noise.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s_noise):
                big.add_(1.0)
            s_other.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s_other):
                o = sweep(t2, a2, ln2, et[:16])
        Vt, E = sweep(t, a, ln, et)
        torch.cuda.synchronize()
        if first is None:
            first = (Vt.clone(), E.clone())
        assert torch.equal(Vt, first[0]) and torch.equal(E, first[1]), it
        if it >= 20:
            if other_first is None:
                other_first = (o[0].clone(), o[1].clone())
            assert torch.equal(o[0], other_first[0]) and torch.equal(o[1], other_first[1]), it
    assert eng.check_device()[0] == 0
    # the same batch through the experiments build with one workgroup per pair (throughput kernels, like the parts)
    exp = _exp_lib()
    stream = torch.cuda.current_stream().cuda_stream
    exp.sdp_set_debug(64)
    st = torch.empty(exp.sdp_state_bytes(B, N, M) // 4, device="cuda")
    vt = torch.empty(B, device="cuda")
    E1 = torch.empty(B, N, M, device="cuda")
    assert exp.sdp_forward_f32(t.data_ptr(), a.data_ptr(), st.data_ptr(), vt.data_ptr(), B, N, M, ln.data_ptr(), 4 << 12, 0, stream) == 0
    assert exp.sdp_backward_f32(et.data_ptr(), st.data_ptr(), E1.data_ptr(), B, N, M, ln.data_ptr(), 4 << 12, 0, stream) == 0
    exp.sdp_set_debug(0)
    torch.cuda.synchronize()
    assert torch.equal(vt, first[0]) and torch.equal(E1, first[1])


def test_sweeps_can_be_captured_in_a_graph():
    """after sdp_init the entry points only enqueue (a memset and up to two kernels per sweep) and can be
    captured.  Forward + backward of an equal-length batch and of a batch with per-pair lengths whose long pairs are spread
    over several workgroups are captured once, replayed on new inputs, and must give the bits of the eager calls."""
    from deepblast_amd._engine import get_engine
    eng = get_engine()
    eng.init()
    for (B, N, M, use_lens) in ((24, 200, 264, False), (96, 704, 512, True)):
        theta, A = datagen.theta_A(94000 + N, B, N, M)
        theta2, _ = datagen.theta_A(94001 + N, B, N, M)
        ln = None
        if use_lens:
            lens = datagen.lengths(94002, B, 1, N)
            lens[:, 1] = np.minimum(lens[:, 1] * M // N + 1, M)
            lens[0] = (N, M)
            ln = torch.from_numpy(lens).cuda()
            assert eng.lib.sdp_plan_parts(0, B, N, M, 1, 0, 256) == 4
        t, a = torch.from_numpy(theta).cuda(), torch.from_numpy(A).cuda()
        et = torch.ones(B, device="cuda")

        def sweep():
            Vt, Q = eng.forward(t, a, 0, ln)
            return Vt, eng.backward(et, Q, (B, N, M), 0, ln)

        sweep()   # warm-up outside the capture (allocator, lazy module loads)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = sweep()
        for src in (theta, theta2):
            t.copy_(torch.from_numpy(src))
            g.replay()
            torch.cuda.synchronize()
            got = (out[0].clone(), out[1].clone())
            ref = sweep()
            torch.cuda.synchronize()
            assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), (B, N, M, use_lens)
    assert eng.check_device()[0] == 0


def test_first_launch_of_a_process_can_be_a_captured_one():
    """sdp_init does everything a first launch would otherwise do lazily (status words, the LDS limit of every kernel build,
    the parts instantiations included): in a FRESH process, init -> capture -> replay, with no eager launch before the
    capture, gives the eager results."""
    import subprocess
    import sys
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, "tests")
import datagen
from deepblast_amd._engine import get_engine
eng = get_engine(); eng.init()
B, N, M = 96, 704, 512
theta, A = datagen.theta_A(95001, B, N, M)
lens = datagen.lengths(95002, B, 1, N); lens[:, 1] = np.minimum(lens[:, 1] * M // N + 1, M); lens[0] = (N, M)
t, a, ln = torch.from_numpy(theta).cuda(), torch.from_numpy(A).cuda(), torch.from_numpy(lens).cuda()
et = torch.ones(B, device="cuda")
def sweep():
    Vt, Q = eng.forward(t, a, 0, ln)
    return Vt, eng.backward(et, Q, (B, N, M), 0, ln)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = sweep()
g.replay(); torch.cuda.synchronize()
got = (out[0].clone(), out[1].clone())
ref = sweep(); torch.cuda.synchronize()
assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
assert eng.check_device()[0] == 0
print("captured-first-launch ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "captured-first-launch ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_exact_zero_chunks_are_skipped_without_changing_a_bit():
    """The fp32 backward sweep does not run the steps of a chunk whose carries, boundary values and cotangent are all
    +0 (sdp_kernels.hip, "exact zeros").  Experiments build, debug bit 4096 runs them all the same: E must be equal
    bit for bit -- signs of zeros included -- on soft scores, on peaked ones (almost everything is zero), with zero and
    negative cotangents (-0 is never skipped), Smith-Waterman, per-pair lengths, both state forms and pairs spread
    over several workgroups."""
    lib = _exp_lib()
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(0).cuda_stream
    cases = [(6, 512, 512, 0, 1.0, False), (3, 700, 330, 1, 8.0, False), (5, 300, 900, 0, 30.0, True), (2, 1100, 1030, 0, 4.0, True),
             (20, 1024, 1024, 0, 8.0, False)]
    nzero = 0
    for ci, (B, N, M, variant, steep, use_lens) in enumerate(cases):
        theta, A = datagen.theta_A(4400 + ci, B, N, M)
        theta *= steep
        et_np = np.ones(B, np.float32)
        et_np[0] = -2.5
        if B > 2:
            et_np[1] = 0.0
            et_np[2] = -0.0
        lens = None
        if use_lens:
            lens = torch.from_numpy(np.minimum(datagen.lengths(4500 + ci, B, 1, N), np.array([N, M])).astype(np.int32)).to(dev)
        t, a, et = torch.from_numpy(theta).to(dev), torch.from_numpy(A).to(dev), torch.from_numpy(et_np).to(dev)
        lp = None if lens is None else lens.data_ptr()
        for exact in (0, 0x100):
            nbytes = lib.sdp_state_bytes_v(B, N, M, variant | exact)
            state = torch.empty(nbytes // 4 + 1, dtype=torch.float32, device=dev)
            vt = torch.empty(B, dtype=torch.float32, device=dev)
            assert lib.sdp_forward_f32(t.data_ptr(), a.data_ptr(), state.data_ptr(), vt.data_ptr(), B, N, M, lp, variant | exact, 0, stream) == 0, lib.sdp_last_error_string()
            outs = []
            for mask in (0, 4096):
                lib.sdp_set_debug(mask)
                try:
                    E = torch.full((B, N, M), float("nan"), device=dev)
                    assert lib.sdp_backward_f32(et.data_ptr(), state.data_ptr(), E.data_ptr(), B, N, M, lp, variant | exact, 0, stream) == 0, lib.sdp_last_error_string()
                    torch.cuda.synchronize()
                finally:
                    lib.sdp_set_debug(0)
                outs.append(E.view(torch.int32).cpu().numpy())
            assert np.array_equal(outs[0], outs[1]), (B, N, M, variant, exact)
            nzero += int((outs[0] == 0).sum())
    assert nzero > 0


def test_zero_chunks_of_the_adjoint_backward_sweep_do_not_change_a_bit():
    """Round 5: the adjoint backward sweep does not run chunks over which E, its carries and its boundary values are all zero,
    nor reads their Q / Qd rows (sdp_kernels.hip, ZSKIP_A).  `variant | SDP_NO_ZERO_SKIP` (the shipped library's control flag)
    runs them all: Ed must be equal as BIT PATTERNS -- the rule for signed zeros: Ed is stored as (float)ed + 0.0f, so no -0
    ever leaves the sweep, whichever way a zero came about -- on soft and steep scores, with negative / zero / -0 cotangents of
    the first-order sweep (E = -0 cells), Smith-Waterman, per-pair lengths, rectangular shapes and many strips; and the same
    flag on the fp32 backward sweep of the shipped library (E bit-identical)."""
    from deepblast_amd import _lib
    from deepblast_amd._engine import get_engine
    lib = get_engine().lib
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(0).cuda_stream
    NOSKIP = _lib.SDP_NO_ZERO_SKIP
    cases = [(6, 512, 512, 0, 1.0, False), (3, 700, 330, 1, 8.0, False), (5, 300, 900, 0, 30.0, True), (2, 1100, 1030, 0, 4.0, True),
             (9, 1024, 1024, 0, 8.0, False), (40, 130, 77, 1, 2.0, True)]
    nzero = 0
    for ci, (B, N, M, variant, steep, use_lens) in enumerate(cases):
        theta, A = datagen.theta_A(5400 + ci, B, N, M)
        theta *= steep
        et_np = np.ones(B, np.float32)
        et_np[0] = -2.5
        if B > 2:
            et_np[1] = 0.0
            et_np[2] = -0.0
        Z = datagen.normal(5500 + ci, (B, N, M))
        lens = None
        if use_lens:
            lens = torch.from_numpy(np.minimum(datagen.lengths(5600 + ci, B, 1, N), np.array([N, M])).astype(np.int32)).to(dev)
        t, a, et, z = (torch.from_numpy(x).to(dev) for x in (theta, A, et_np, Z))
        lp = None if lens is None else lens.data_ptr()
        X = 0x100 | variant
        state = torch.empty(lib.sdp_state_d_bytes(B, N, M) // 4 + 1, dtype=torch.float32, device=dev)
        state_d = torch.empty_like(state)
        vt, vtd = torch.empty(B, device=dev), torch.empty(B, device=dev)
        assert lib.sdp_forward_f32(t.data_ptr(), a.data_ptr(), state.data_ptr(), vt.data_ptr(), B, N, M, lp, X, 0, stream) == 0, lib.sdp_last_error_string()
        Es = []
        for flag in (0, NOSKIP):
            E = torch.full((B, N, M), float("nan"), device=dev)
            assert lib.sdp_backward_f32(et.data_ptr(), state.data_ptr(), E.data_ptr(), B, N, M, lp, X | flag, 0, stream) == 0, lib.sdp_last_error_string()
            Es.append(E)
        torch.cuda.synchronize()
        assert np.array_equal(Es[0].view(torch.int32).cpu().numpy(), Es[1].view(torch.int32).cpu().numpy()), ("E", B, N, M)
        assert lib.sdp_adjoint_forward_f32(state.data_ptr(), z.data_ptr(), None, vtd.data_ptr(), state_d.data_ptr(), B, N, M, lp, variant, 0, stream) == 0
        outs = []
        for flag in (0, NOSKIP):
            Ed = torch.full((B, N, M), float("nan"), device=dev)
            assert lib.sdp_adjoint_backward_f32(Es[0].data_ptr(), state.data_ptr(), state_d.data_ptr(), Ed.data_ptr(), B, N, M, lp, variant | flag, 0, stream) == 0, lib.sdp_last_error_string()
            torch.cuda.synchronize()
            outs.append(Ed.view(torch.int32).cpu().numpy())
        assert np.array_equal(outs[0], outs[1]), ("Ed", B, N, M, variant, int((outs[0] != outs[1]).sum()))
        assert not (outs[0] == np.int32(-2**31)).any()      # no -0 in Ed
        nzero += int((outs[0] == 0).sum())
    assert nzero > 0


def test_the_two_builds_of_the_packed_backward_sweep_agree_bit_for_bit():
    """Round 5: the packed backward sweep exists twice -- `sdp_bwd_kernel` and `sdp_bwd_pipe_kernel` (the chunk as one software
    pipeline) -- and the launch plan takes the second only where every CU holds one pair of long rows (sdp_api.hip: bwd_pipe_pays).
    Which build a pair meets depends on the BATCH it arrives in, so the two must agree as bit patterns: the same pairs swept as one
    batch of 256 (pipelined build) and as two batches of 128 (the other one), soft and steep scores, NW and SW, and with the
    zero-chunk skip switched off."""
    from deepblast_amd import _lib
    from deepblast_amd._engine import get_engine
    lib = get_engine().lib
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(0).cuda_stream
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    B, N, M = cus, 512, 1024

    def plan_id(b):
        kid, chunk, waves, lds = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
        assert lib.sdp_plan(1, b, N, M, 0, 0, cus, ctypes.byref(kid), ctypes.byref(chunk), ctypes.byref(waves), ctypes.byref(lds)) == 0
        return kid.value
    assert plan_id(B) == 36 and plan_id(B // 2) == 1, (plan_id(B), plan_id(B // 2))
    for ci, (variant, steep, flag) in enumerate([(0, 1.0, 0), (1, 6.0, 0), (0, 3.0, _lib.SDP_NO_ZERO_SKIP)]):
        th, A = datagen.theta_A(5900 + ci, 32, N, M)
        th = th * np.float32(steep)
        t = torch.from_numpy(np.tile(th, (B // 32, 1, 1))).to(dev)
        a = torch.from_numpy(np.tile(A, (B // 32, 1, 1))).to(dev)
        et = torch.from_numpy((datagen.uniform(5950 + ci, (B,)) + np.float32(0.5))).to(dev)
        E1 = torch.full((B, N, M), float("nan"), device=dev)
        E2 = torch.full((B, N, M), float("nan"), device=dev)
        vt1, vt2 = torch.empty(B, device=dev), torch.empty(B, device=dev)
        st = torch.empty(lib.sdp_state_bytes(B, N, M) // 4 + 1, dtype=torch.float32, device=dev)
        assert lib.sdp_forward_f32(t.data_ptr(), a.data_ptr(), st.data_ptr(), vt1.data_ptr(), B, N, M, None, variant, 0, stream) == 0, lib.sdp_last_error_string()
        assert lib.sdp_backward_f32(et.data_ptr(), st.data_ptr(), E1.data_ptr(), B, N, M, None, variant | flag, 0, stream) == 0, lib.sdp_last_error_string()
        h = B // 2
        for lo in (0, h):
            assert lib.sdp_forward_f32(t[lo:].data_ptr(), a[lo:].data_ptr(), st.data_ptr(), vt2[lo:].data_ptr(), h, N, M, None, variant, 0, stream) == 0
            assert lib.sdp_backward_f32(et[lo:].data_ptr(), st.data_ptr(), E2[lo:].data_ptr(), h, N, M, None, variant | flag, 0, stream) == 0
        torch.cuda.synchronize()
        assert torch.equal(vt1.view(torch.int32), vt2.view(torch.int32)), ci
        assert torch.equal(E1.view(torch.int32), E2.view(torch.int32)), (ci, int((E1.view(torch.int32) != E2.view(torch.int32)).sum()))
        assert bool(torch.isfinite(E1).all())


@pytest.mark.parametrize("case", [(3, 150, 200, True, 0), (3, 150, 200, True, 0x100), (80, 192, 256, True, 0), (80, 130, 250, False, 0), (2, 100, 331, False, 0),
                                  (80, 192, 256, True, 0x100)],
                         ids=["lat-lens", "lat-lens-exact", "tp-lens", "tp-gen-partial", "lat-partial", "tp-lens-exact"])
def test_what_lies_beside_the_matrix_takes_no_part_in_anything(case):
    """VERDICT r5 2c.  The forward sweep's windowed form tests the range of EVERY lane's inputs and values, also of lanes whose
    cell lies outside the matrix -- and until round 6 those lanes computed from whatever the loaded groups brought along: up to
    three floats of the neighbouring row or pair, the padding of a batch with per-pair lengths, the rows below a partial strip,
    the memory in front of and behind the tensor.  Results never depended on it, but WHICH FORM a block ran in did (timing, and
    the last bits of the packed state), which is how round 5's wrong-result bug hid behind a plane offset of three floats.
    Here everything beside the matrices is poisoned (NaN, +-inf, +-1e30) in one run and zero in the other, at every plane offset:
    Vt and E must be equal as bit patterns AND every traced block must have run in the same form (experiments build: the form
    codes of pair 0's first 40 blocks of every strip)."""
    lib = _exp_lib()
    lib.sdp_set_trace.restype, lib.sdp_set_trace.argtypes = ctypes.c_int, [ctypes.c_void_p]
    B, N, M, use_lens, xflag = case
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(0).cuda_stream
    rng = np.random.default_rng(6006 + N + M)
    theta, A = datagen.theta_A(6100 + N, B, N, M)
    theta = (theta * 3.0).astype(np.float32)   # (steeper than the benchmark's scores: some blocks near the window's edge)
    lens = None
    if use_lens:
        lens = np.stack([rng.integers(1, N + 1, B), rng.integers(1, M + 1, B)], axis=1).astype(np.int32)
        lens[0] = (N - 3, M - 5)            # the traced pair: ragged in both directions
        lens[1 % B] = (N, M)
    poison = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30], np.float32)
    PAD = 4096
    results = []
    for variant in (0, 1):
        per_variant = []
        for offset in range(4):
            runs = []
            for dirty in (False, True):
                bufs = []
                for src in (theta, A):
                    fill = poison[rng.integers(0, 5, src.size + 2 * PAD)] if dirty else np.zeros(src.size + 2 * PAD, np.float32)
                    buf = torch.from_numpy(fill.astype(np.float32)).to(dev)
                    view = buf[PAD + offset:PAD + offset + src.size].view(B, N, M)
                    x = src.copy()
                    if lens is not None:   # the padding of every pair as well
                        for b in range(B):
                            pad = poison[rng.integers(0, 5, (N, M))] if dirty else np.zeros((N, M), np.float32)
                            x[b, lens[b, 0]:, :] = pad[lens[b, 0]:, :]
                            x[b, :, lens[b, 1]:] = pad[:, lens[b, 1]:]
                    view.copy_(torch.from_numpy(x))
                    bufs.append((buf, view))
                t, a = bufs[0][1], bufs[1][1]
                ln = None if lens is None else torch.from_numpy(lens).to(dev)
                lp = None if ln is None else ln.data_ptr()
                nb = max(lib.sdp_state_bytes(B, N, M), lib.sdp_state_d_bytes(B, N, M))
                st = torch.zeros(nb // 4 + 64, device=dev)
                vt = torch.empty(B, device=dev)
                E = torch.full((B, N, M), 7.0, device=dev)
                et = torch.ones(B, device=dev)
                trace = torch.zeros(4 * 4 * 4 * 40 * 8, dtype=torch.int64, device=dev)
                lib.sdp_set_trace(trace.data_ptr())
                try:
                    assert lib.sdp_forward_f32(t.data_ptr(), a.data_ptr(), st.data_ptr(), vt.data_ptr(), B, N, M, lp, variant | xflag, 0, stream) == 0
                    torch.cuda.synchronize()
                finally:
                    lib.sdp_set_trace(None)
                assert lib.sdp_backward_f32(et.data_ptr(), st.data_ptr(), E.data_ptr(), B, N, M, lp, variant | xflag, 0, stream) == 0
                torch.cuda.synchronize()
                forms = trace.cpu().numpy().reshape(4, 4, 4, 40, 8)[0, :, :, :, 6].copy()   # pair 0: [wave][strip round][block]
                runs.append((vt.cpu().numpy().view(np.uint32), E.cpu().numpy().view(np.uint32), forms))
            clean, dirty_run = runs
            assert np.array_equal(clean[0], dirty_run[0]), (variant, offset, "Vt")
            assert np.array_equal(clean[1], dirty_run[1]), (variant, offset, "E", int((clean[1] != dirty_run[1]).sum()))
            assert (clean[2] > 0).sum() >= 4, "no traced blocks: the trace did not see pair 0"
            assert np.array_equal(clean[2], dirty_run[2]), (variant, offset, "forms", np.argwhere(clean[2] != dirty_run[2])[:5].tolist())
            per_variant.append(clean)
        # ... and with per-pair lengths the plane offset itself changes nothing either (the same problem at four alignments: the
        # general-pitch build cuts its groups elsewhere, but everything outside a pair's block is zero for every build).  Without
        # lengths the groups of the general-pitch and latency builds straddle into the pair's own neighbouring rows -- the same
        # on every run, but not the same for every build: there only Vt and E at the ordinary bound are held across offsets.
        for k in range(1, 4):
            if lens is not None:
                assert np.array_equal(per_variant[0][0], per_variant[k][0]) and np.array_equal(per_variant[0][1], per_variant[k][1]), (variant, k)
                assert np.array_equal(per_variant[0][2], per_variant[k][2]), (variant, k, "forms by offset")
            else:
                assert parity.abs_err(per_variant[k][1].view(np.float32), per_variant[0][1].view(np.float32)) <= 1e-5, (variant, k)
        results.append(per_variant[0])
    # against the oracle once (NW), so that "equal" is not "equally wrong"
    ref = (parity.oracle_lens(theta, A, None, None, 0, lens) if lens is not None else parity.oracle_all(theta, A, None, None, 0))
    assert parity.rel_err(results[0][0].view(np.float32), ref["Vt"]) <= parity.TOL
    assert parity.abs_err(results[0][1].view(np.float32), ref["E"]) <= parity.TOL
