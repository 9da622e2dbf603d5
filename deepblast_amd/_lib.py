"""ctypes binding of libsdp_hip.so (the C ABI in include/sdp.h).

This is the whole FFI: raw device pointers, ints and a stream handle.  There is no
CPU fallback -- if the shared library is missing the import fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SDP_LIB_PATH lets experiments (tools/gpu_tune.py variants) run the whole test-suite on another build
LIB_PATH = os.environ.get("SDP_LIB_PATH") or os.path.join(_HERE, "libsdp_hip.so")

SDP_NW, SDP_SW = 0, 1
SDP_NO_ZERO_SKIP, SDP_NO_FILL = 0x800, 0x10000   # include/sdp.h: flags of the backward sweeps

_c_f32p = ctypes.c_void_p
_c_i32p = ctypes.c_void_p

# name -> (restype, argtypes); must list every symbol include/sdp.h declares
SIGNATURES = {
    "sdp_version": (ctypes.c_int, []),
    "sdp_last_error_string": (ctypes.c_char_p, []),
    "sdp_max_cols": (ctypes.c_int, []),
    "sdp_state_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "sdp_state_d_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "sdp_state_bytes_v": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "sdp_state_d_bytes_v": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "sdp_plan": (ctypes.c_int, [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_int)] * 3 + [ctypes.POINTER(ctypes.c_size_t)]),
    "sdp_plan_parts": (ctypes.c_int, [ctypes.c_int] * 7),
    "sdp_forward_f32": (ctypes.c_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, _c_i32p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "sdp_backward_f32": (ctypes.c_int, [_c_f32p, _c_f32p, _c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        _c_i32p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "sdp_state_pair_stride": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "sdp_backward_range_f32": (ctypes.c_int, [_c_f32p, _c_f32p, _c_f32p] + [ctypes.c_int] * 7 + [ctypes.c_void_p]),
    "sdp_adjoint_forward_f32": (ctypes.c_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_int, _c_i32p, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_void_p]),
    "sdp_adjoint_forward_loss_f32": (ctypes.c_int, [_c_f32p] * 5 + [ctypes.c_int] + [_c_f32p] * 2 + [ctypes.c_int] * 3 +
                                     [_c_i32p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "sdp_adjoint_backward_f32": (ctypes.c_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, _c_i32p, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_void_p]),
    "sdp_state_bytes_f64": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "sdp_forward_f64": (ctypes.c_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, _c_i32p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "sdp_backward_f64": (ctypes.c_int, [_c_f32p, _c_f32p, _c_f32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        _c_i32p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "sdp_adjoint_forward_f64": (ctypes.c_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_int, _c_i32p, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_void_p]),
    "sdp_adjoint_backward_f64": (ctypes.c_int, [_c_f32p, _c_f32p, _c_f32p, _c_f32p, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, _c_i32p, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_void_p]),
    "sdp_scores_f32": (ctypes.c_int, [_c_f32p] * 6 + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "sdp_scores_backward_ws_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "sdp_scores_backward_f32": (ctypes.c_int, [_c_f32p] * 13 + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "sdp_traceback_capacity": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "sdp_traceback_i32": (ctypes.c_int, [_c_f32p, _c_i32p, _c_i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_i32p,
                                         ctypes.c_int, ctypes.c_void_p]),
    "sdp_traceback_rule_i32": (ctypes.c_int, [_c_f32p, _c_i32p, _c_i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int, _c_i32p,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "sdp_init": (ctypes.c_int, [ctypes.c_int]),
    "sdp_loss_forward_f32": (ctypes.c_int, [_c_f32p, _c_f32p, _c_f32p, _c_i32p, ctypes.c_void_p, _c_i32p, ctypes.c_int,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "sdp_loss_backward_f32": (ctypes.c_int, [_c_f32p, _c_f32p, _c_f32p, _c_i32p, _c_f32p, _c_f32p, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "sdp_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "sdp_comm_init": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]),
    "sdp_comm_all_gather_f32": (ctypes.c_int, [ctypes.c_void_p, _c_f32p, _c_f32p, ctypes.c_size_t, ctypes.c_void_p]),
    "sdp_comm_destroy": (ctypes.c_int, [ctypes.c_void_p]),
    "sdp_comm_last_error_string": (ctypes.c_char_p, []),
    "sdp_selftest": (ctypes.c_int, [ctypes.c_int]),
    "sdp_device_status": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_int32)]),
}
# only in -DSDP_EXPERIMENTS builds (deepblast_amd/libsdp_hip_exp.so; never the shipped library)
EXPERIMENT_SIGNATURES = {
    "sdp_set_debug": (ctypes.c_int, [ctypes.c_int]),
    "sdp_set_trace": (ctypes.c_int, [ctypes.c_void_p]),
}


def SDP_WAVES(w):
    """include/sdp.h: or-ed into `variant`, run with w wavefronts per pair."""
    return (int(w) & 0xf) << 12

_LIB = None


class SdpLibraryMissing(ImportError):
    pass


class HandoffTimeout(RuntimeError):
    """SDP_E_HANDOFF: a kernel of an earlier launch gave up waiting for a strip hand-off; that launch's results
    are invalid (include/sdp.h)."""


def load():
    """Load (once) and return the ctypes handle; raise if the HIP library was not built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SdpLibraryMissing(
                f"{LIB_PATH} not found: the HIP engine is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or deepblast_amd/build.py). "
                "deepblast_amd has no CPU fallback.")
        _LIB = load_path(LIB_PATH)
    return _LIB


def load_path(path):
    """A fresh, bound handle of the library at `path` (tests load the -DSDP_EXPERIMENTS build next to the shipped one)."""
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in EXPERIMENT_SIGNATURES.items():
        if hasattr(lib, name):
            getattr(lib, name).restype = res
            getattr(lib, name).argtypes = args
    return lib


def check(rc, what):
    """Map the ABI's status codes onto Python exceptions."""
    if rc == 0:
        return
    msg = load().sdp_last_error_string().decode("utf-8", "replace")
    if rc == -3:
        raise ValueError(f"{what}: {msg}")
    if rc in (-1, -2, -4, -5):
        raise ValueError(f"{what}: {msg}")
    if rc == -7:
        raise HandoffTimeout(f"{what}: {msg}")
    raise RuntimeError(f"{what}: {msg} (status {rc})")
