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


def _declared():
    src = open(os.path.join(ROOT, "include", "sdp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sdp_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from deepblast_amd import _lib
    names = _declared()
    assert len(names) >= 10
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sdp.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_limits(lib):
    assert lib.sdp_version() == 100
    assert lib.sdp_max_cols() == 2048  # reference GPU path: max_cols = 2048 (nw_cuda.py:11)


def test_state_bytes(lib):
    # Q: 6 B (two 23-bit weights) per cell of the skewed layout; Qd: float2 per cell
    assert lib.sdp_state_bytes(256, 512, 512) == 256 * 8 * 576 * 64 * 6
    assert lib.sdp_state_bytes(1, 1, 1) == 1 * 1 * 64 * 64 * 6
    assert lib.sdp_state_bytes(3, 65, 2) == 3 * 2 * 128 * 64 * 6
    assert lib.sdp_state_d_bytes(256, 512, 512) == 256 * 8 * 576 * 64 * 8
    assert lib.sdp_state_d_bytes(0, 5, 5) == 0
    assert lib.sdp_state_bytes(0, 5, 5) == 0


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
    assert lib.sdp_set_waves(9, 1) == -1
