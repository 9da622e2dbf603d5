"""CPU: the C-ABI library loads without a GPU and exports exactly what include/sdp.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from deepblast_amd import _lib, build
    build.build()  # hipcc cross-compiles for gfx950 without a GPU
    return _lib.load()


def _declared(experiments=False):
    src = open(os.path.join(ROOT, "include", "sdp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    exp = re.findall(r"#ifdef SDP_EXPERIMENTS(.*?)#endif", src, flags=re.S)
    if experiments:
        src = "".join(exp)
    else:
        src = re.sub(r"#ifdef SDP_EXPERIMENTS.*?#endif", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sdp_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from deepblast_amd import _lib
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sdp.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names
    # the shipped library has no experiment switches (no wrong-results knob reachable through the ABI)
    for n in _declared(experiments=True):
        assert n in _lib.EXPERIMENT_SIGNATURES and not hasattr(lib, n), n
    for gone in ("sdp_set_waves", "sdp_probe"):
        assert not hasattr(lib, gone)


def test_version_and_limits(lib):
    assert lib.sdp_version() == 106
    assert lib.sdp_max_cols() == 2048  # reference GPU path: max_cols = 2048 (nw_cuda.py:11)


def test_state_bytes(lib):
    # Q: 5 B (two 20-bit weights) per cell of the skewed, padded layout; Qd: float2 per cell (+ a tail for the launch order of
    # variable-length batches: B ints, 256-byte granules)
    # ... and, for pairs of more than four strips, the bridge rows between the parts a pair may be cut into: per pair
    # (ceil(strips / 4) - 1) rows of roundup(M + 63, 64) + 64 granules of 8 bytes
    bridge = 256 * 1 * (576 + 64 + 40) * 8 + 256 * 2 * 4   # (+ the dispatch order of the parts: one int per workgroup)
    assert lib.sdp_state_bytes(256, 512, 512) == 256 * 8 * 576 * 64 * 5 + 1024 + bridge
    assert lib.sdp_state_bytes(1, 1, 1) == 1 * 1 * 64 * 64 * 5 + 256
    assert lib.sdp_state_bytes(3, 65, 2) == 3 * 2 * 128 * 64 * 5 + 256
    assert lib.sdp_state_bytes(2, 1024, 100) == 2 * 16 * 192 * 64 * 5 + 256 + 2 * 3 * (192 + 64 + 16) * 8 + 256
    assert lib.sdp_state_d_bytes(256, 512, 512) == 256 * 8 * 576 * 64 * 8 + 1024 + bridge
    assert lib.sdp_state_d_bytes(0, 5, 5) == 0
    assert lib.sdp_state_bytes(0, 5, 5) == 0


def test_header_states_the_packed_state_format_the_library_uses(lib):
    """include/sdp.h says how many bytes per cell the packed state takes (SDP_PACKED_STATE_BYTES_PER_CELL and the prose
    next to it): the number must be the one sdp_state_bytes is built on, so that the header cannot rot again (round 4: the
    header still said 6 bytes / 23-bit fields after the format had become 5 bytes / 20-bit)."""
    import re
    hdr = open(os.path.join(ROOT, "include", "sdp.h")).read()
    m = re.search(r"#define\s+SDP_PACKED_STATE_BYTES_PER_CELL\s+(\d+)", hdr)
    assert m, "include/sdp.h must define SDP_PACKED_STATE_BYTES_PER_CELL"
    per_cell = int(m.group(1))
    # (B, N, M) = (1, 64, 2): one strip, 2 + 63 -> 128 steps of 64 lanes; the tail (launch order) is 256 bytes
    assert lib.sdp_state_bytes(1, 64, 2) == 128 * 64 * per_cell + 256
    # two pairs more of 512 x 512 cost two records of 8 strips x 576 steps x 64 lanes (+ 1024 bytes of dispatch map for the
    # parts of the two extra pairs): the per-cell figure again, through a difference that drops the rest of the tail
    assert lib.sdp_state_pair_stride(1024, 1024, 0) == 16 * 1088 * 64 * per_cell
    assert lib.sdp_state_pair_stride(512, 512, 0) == 8 * 576 * 64 * per_cell
    assert f"{per_cell} bytes" in hdr and "20-bit" in hdr and "23-bit" not in hdr and "6 bytes" not in hdr
    for doc in ("INTEGRATION.md", os.path.join("deepblast_amd", "_engine.py"), os.path.join("deepblast_amd", "_dp.py")):
        text = open(os.path.join(ROOT, doc)).read()
        assert "6 bytes per cell" not in text and "6 B/cell" not in text, doc


def test_argument_errors_need_no_gpu(lib):
    one = ctypes.c_void_p(16)
    assert lib.sdp_forward_f32(None, one, one, one, 1, 1, 1, None, 0, 0, None) == -1
    assert b"null" in lib.sdp_last_error_string()
    assert lib.sdp_forward_f32(one, one, one, one, 0, 1, 1, None, 0, 0, None) == -2
    assert lib.sdp_forward_f32(one, one, one, one, 1, 1, 2049, None, 0, 0, None) == -3
    assert lib.sdp_forward_f32(one, one, one, one, 1, 1, 1, None, 7, 0, None) == -4
    assert lib.sdp_backward_f32(one, one, None, 1, 1, 1, None, 0, 0, None) == -1
    assert lib.sdp_adjoint_forward_f32(one, None, None, one, one, 1, 1, 1, None, 0, 0, None) == -1
    assert lib.sdp_adjoint_backward_f32(one, one, one, one, 1, 1, 1 << 20, None, 0, 0, None) == -3
    # flags in `variant` (SDP_EXACT_STATE, SDP_WAVES) are stripped before validation; junk above them is not
    assert lib.sdp_forward_f32(one, one, one, one, 0, 1, 1, None, 0x100 | (8 << 12), 0, None) == -2
    assert lib.sdp_forward_f32(one, one, one, one, 1, 1, 1, None, 1 << 20, 0, None) == -4
    info = (ctypes.c_int32 * 4)()
    assert lib.sdp_device_status(0, info) == 0 and list(info) == [0, 0, 0, 0]   # nothing launched: no status block


def _plan(lib, pass_, B, N, M, lens=0, exact=0, cus=256):
    import ctypes
    kid, chunk, waves, lds = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
    assert lib.sdp_plan(pass_, B, N, M, lens, exact, cus, ctypes.byref(kid), ctypes.byref(chunk), ctypes.byref(waves),
                        ctypes.byref(lds)) == 0
    return kid.value, chunk.value, waves.value, lds.value


def test_thin_long_problems_take_the_exact_state(lib):
    """Round 5: fewer than 32 rows (or columns) with more than 512 on the other axis -- the packed weights' rounding does not
    average out over many paths there (2 x 2048: 1.0e-4) -- use the float2 state, like problems with N + M > 4096; the sizing
    function follows (sdp_api.hip: exact_for).  Round 6: thin long PAIRS of a batch with per-pair lengths are routed to the float2
    build inside their own packed record, which needs more than 64 rows and more than 65 columns of padded shape to fit -- padded
    shapes with min(N, M) < 66 (and max > 512) therefore take the float2 state as a whole."""
    for (N, M, exact) in [(2, 1772, True), (1772, 2, True), (31, 513, True), (32, 2048, True), (65, 2048, True), (66, 2048, False), (2048, 65, True), (2048, 66, False),
                          (31, 512, False), (64, 512, False), (64, 960, True), (7, 1361, True),
                          (2048, 2048, False), (2049, 2048, True), (512, 512, False)]:
        assert (lib.sdp_state_bytes(3, N, M) == lib.sdp_state_d_bytes(3, N, M)) == exact, (N, M)


def test_launch_plan_policy(lib):
    """Which build / how many waves a launch uses, checked without a device."""
    LDS = 160 * 1024
    # headline: one pair per CU -> throughput builds, 4 waves, long chunks
    # (ids 29 .. 35 would be the builds of the 18-bit packed state, -DSDP_Q18=1: measured in round 5 and not adopted)
    assert _plan(lib, 0, 256, 512, 512)[:3] == (0, 32, 4)
    assert _plan(lib, 1, 256, 512, 512)[:3] == (1, 32, 4)
    # small batch -> latency builds, 8 waves, short chunks
    assert _plan(lib, 0, 16, 512, 512)[:3] == (6, 16, 8)
    # (backward sweep, round 5: the K = 32 packed build with 8 waves -- it needs < 256 registers now; the exact state keeps K = 16)
    assert _plan(lib, 1, 16, 512, 512)[:3] == (1, 32, 8) and _plan(lib, 1, 100, 512, 512)[:3] == (1, 32, 8)
    # the packed backward sweep's pipelined twin (id 36): only where every CU holds one pair of long rows (steady-state table, sdp_api.hip)
    assert _plan(lib, 1, 256, 512, 512)[:3] == (1, 32, 4) and _plan(lib, 1, 256, 1024, 1024)[:3] == (36, 32, 4)
    assert _plan(lib, 1, 256, 512, 1024)[0] == 36 and _plan(lib, 1, 256, 768, 640)[0] == 1 and _plan(lib, 1, 128, 1024, 1024)[:3] == (1, 32, 4)
    assert _plan(lib, 1, 1024, 512, 512)[:3] == (1, 32, 2) and _plan(lib, 1, 384, 1024, 1024)[0] == 1 and _plan(lib, 1, 256, 2048, 256)[0] == 1
    # (forward sweep: the throughput build from ~72 pairs on)
    assert _plan(lib, 0, 64, 512, 512)[:3] == (6, 16, 8) and _plan(lib, 0, 80, 512, 512)[:3] == (0, 32, 4)
    # a pair spread over several CUs, four strips (one per wave of the throughput builds) per workgroup: where it was
    # measured to pay -- per-pair lengths and a batch within the CU count: forward sweep from three parts on, backward sweep
    # from two; equal pairs: the backward sweep of a few pairs of more than twelve strips; never the adjoint pair
    pp = lambda pass_, B, N, M, lens, exact=0: lib.sdp_plan_parts(pass_, B, N, M, lens, exact, 256)
    # (round 5: the backward sweep no longer takes parts with per-pair lengths -- re-measured, profiles/r05_parts_table.txt)
    assert pp(0, 256, 1022, 1020, 1) == 4 and pp(1, 256, 1022, 1020, 1) == 0 and pp(0, 256, 640, 640, 1) == 4
    assert pp(0, 256, 512, 512, 1) == 0 and pp(1, 256, 512, 512, 1) == 0 and pp(1, 256, 256, 512, 1) == 0
    assert pp(0, 700, 1022, 1020, 1) == 0 and pp(1, 700, 1022, 1020, 1) == 0
    assert pp(0, 16, 1024, 1024, 0) == 0 and pp(1, 16, 1024, 1024, 0) == 4 and pp(1, 64, 640, 500, 0) == 0 and pp(1, 128, 1024, 512, 0) == 0
    assert pp(2, 16, 1024, 1024, 1, 1) == 0 and pp(3, 16, 1024, 1024, 1, 1) == 0
    assert _plan(lib, 1, 16, 1024, 1024)[:3] == (23, 32, 4)   # (the parts instantiation of the throughput build)
    # more pairs than CUs: two waves when that needs fewer rounds
    assert _plan(lib, 0, 512, 512, 512)[2] == 2 and _plan(lib, 0, 768, 512, 512)[2] == 4
    # per-pair lengths on a batch that does not queue up: long pairs in parts (above), otherwise treated like a small batch
    assert _plan(lib, 0, 256, 1024, 1024, lens=1)[:3] == (21, 32, 4)
    # (round 6: with per-pair lengths, or N not a multiple of 64, the forward builds' twins that clean what lies beside the
    #  matrix -- 37 / 38 / 39 / 40 for 0 / 9 / 6 / 5; the aligned no-lengths builds carry none of that code)
    assert _plan(lib, 0, 256, 512, 1024, lens=1)[:3] == (39, 16, 8)
    assert _plan(lib, 0, 600, 512, 512, lens=1)[0] == 37 and _plan(lib, 0, 256, 500, 512)[0] == 37 and _plan(lib, 0, 256, 500, 512, exact=1)[0] == 38
    assert _plan(lib, 0, 16, 500, 512)[0] == 39 and _plan(lib, 0, 16, 500, 512, exact=1)[0] == 40 and _plan(lib, 0, 16, 512, 512, lens=1)[0] == 39
    # exact state for the adjoint sweeps: its own build
    assert _plan(lib, 0, 256, 512, 512, exact=1)[0] == 9 and _plan(lib, 0, 16, 512, 512, exact=1)[0] == 5
    assert _plan(lib, 1, 256, 512, 512, exact=1)[:3] == (7, 32, 4) and _plan(lib, 1, 16, 512, 512, exact=1)[:3] == (8, 16, 8)
    # never more waves than strips; LDS always fits, also at the column limit
    assert _plan(lib, 0, 4, 100, 100)[2] == 2
    for pass_ in range(4):
        for (B, N, M) in [(256, 512, 512), (4, 64, 2048), (300, 2000, 2048), (1, 1, 1)]:
            kid, chunk, waves, lds = _plan(lib, pass_, B, N, M)
            assert 1 <= waves <= 8 and lds <= LDS, (pass_, B, N, M, kid, waves, lds)
    # the long-M fallback: throughput builds do not fit with 2048 columns
    assert _plan(lib, 0, 256, 512, 2048)[0] == 6 and _plan(lib, 1, 256, 512, 2048)[0] in (1, 4, 36)
    assert _plan(lib, 1, 256, 512, 2048, exact=1)[0] in (7, 8)
