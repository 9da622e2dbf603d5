"""ctypes front-end of the CPU oracle (oracle/sdp_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (deepblast_amd) never imports it.

All functions take / return numpy arrays with the reference's own shapes
(deepblast/nw.py:104-116, 342-386): Q (B,N+2,M+2,3), E/Ztheta/Ed (B,N+2,M+2).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}

NW, SW = 0, 1


def build(force=False):
    """Compile liboracle.so / liboracle_omp.so with gcc (a few seconds)."""
    need = force or not all(
        os.path.exists(os.path.join(_HERE, n)) for n in ("liboracle.so", "liboracle_omp.so"))
    if need:
        subprocess.check_call(["make", "-C", _HERE, "-B", "all"],
                              stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def _lib(omp=False):
    name = "liboracle_omp.so" if omp else "liboracle.so"
    if name not in _LIBS:
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        for suf in ("f32", "f64"):
            for fn in ("forward", "backward", "adjoint_forward", "adjoint_backward"):
                getattr(lib, f"oracle_{fn}_{suf}").restype = ctypes.c_int
        _LIBS[name] = lib
    return _LIBS[name]


def _suf(dtype):
    dtype = np.dtype(dtype)
    if dtype == np.float32:
        return "f32"
    if dtype == np.float64:
        return "f64"
    raise TypeError(f"oracle supports float32/float64, got {dtype}")


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def forward(theta, A, variant=NW, omp=False):
    """-> (Vt (B,), Q (B,N+2,M+2,3)) in theta's dtype."""
    dt = theta.dtype
    theta, A = _c(theta, dt), _c(A, dt)
    B, N, M = theta.shape
    Q = np.empty((B, N + 2, M + 2, 3), dtype=dt)
    Vt = np.empty((B,), dtype=dt)
    rc = getattr(_lib(omp), f"oracle_forward_{_suf(dt)}")(
        _p(theta), _p(A), _p(Q), _p(Vt), B, N, M, int(variant))
    if rc:
        raise MemoryError("oracle_forward failed")
    return Vt, Q


def backward(Et, Q, variant=NW, omp=False):
    """-> E (B,N+2,M+2)."""
    dt = Q.dtype
    Et, Q = _c(Et, dt), _c(Q, dt)
    B, N2, M2, _ = Q.shape
    E = np.empty((B, N2, M2), dtype=dt)
    rc = getattr(_lib(omp), f"oracle_backward_{_suf(dt)}")(
        _p(Et), _p(Q), _p(E), B, N2 - 2, M2 - 2, int(variant))
    if rc:
        raise MemoryError("oracle_backward failed")
    return E


def adjoint_forward(Q, Ztheta, ZA, omp=False):
    """-> (Vtd (B,), Qd (B,N+2,M+2,3)).  Ztheta is (B,N+2,M+2), ZA is (B,N,M)."""
    dt = Q.dtype
    Q, Ztheta, ZA = _c(Q, dt), _c(Ztheta, dt), _c(ZA, dt)
    B, N2, M2, _ = Q.shape
    Qd = np.empty_like(Q)
    Vtd = np.empty((B,), dtype=dt)
    rc = getattr(_lib(omp), f"oracle_adjoint_forward_{_suf(dt)}")(
        _p(Q), _p(Ztheta), _p(ZA), _p(Vtd), _p(Qd), B, N2 - 2, M2 - 2)
    if rc:
        raise MemoryError("oracle_adjoint_forward failed")
    return Vtd, Qd


def adjoint_backward(E, Q, Qd, omp=False):
    """-> Ed (B,N+2,M+2)."""
    dt = Q.dtype
    E, Q, Qd = _c(E, dt), _c(Q, dt), _c(Qd, dt)
    B, N2, M2, _ = Q.shape
    Ed = np.empty((B, N2, M2), dtype=dt)
    rc = getattr(_lib(omp), f"oracle_adjoint_backward_{_suf(dt)}")(
        _p(E), _p(Q), _p(Qd), _p(Ed), B, N2 - 2, M2 - 2)
    if rc:
        raise MemoryError("oracle_adjoint_backward failed")
    return Ed


def fwd_bwd(theta, A, Et=None, variant=NW, omp=False):
    """Vt, E[:,1:-1,1:-1] -- what `dec(theta,A).backward(Et)` leaves in theta.grad."""
    Vt, Q = forward(theta, A, variant, omp)
    if Et is None:
        Et = np.ones_like(Vt)
    E = backward(Et, Q, variant, omp)
    return Vt, np.ascontiguousarray(E[:, 1:-1, 1:-1]), Q, E


def double_backward(Q, E, Ztheta_inner, ZA=None, omp=False):
    """Ed[:,1:-1,1:-1], Vtd for a cotangent Ztheta_inner (B,N,M) on theta.grad
    (deepblast/nw.py:357-386: autograd pads the cotangent of the sliced E with zeros)."""
    B, N2, M2, _ = Q.shape
    Zt = np.zeros((B, N2, M2), dtype=Q.dtype)
    Zt[:, 1:-1, 1:-1] = Ztheta_inner
    if ZA is None:
        ZA = np.zeros((B, N2 - 2, M2 - 2), dtype=Q.dtype)
    Vtd, Qd = adjoint_forward(Q, Zt, ZA, omp)
    Ed = adjoint_backward(E, Q, Qd, omp)
    return np.ascontiguousarray(Ed[:, 1:-1, 1:-1]), Vtd, Qd
