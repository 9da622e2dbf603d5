"""CPU: pin the oracle (oracle/sdp_oracle.c) to the fixtures generated from the real
reference (oracle/gen_golden.py -> tests/golden/).  fp32 tensors must match bit-for-bit up to
a final-ulp libm difference; fp64 to 1e-12."""
import os

import numpy as np
import pytest

import datagen
from oracle import oracle

VAR = {"nw": oracle.NW, "sw": oracle.SW}


def _close(got, ref, what):
    tol = 2e-7 if ref.dtype == np.float32 else 1e-12
    scale = np.maximum(1.0, np.abs(ref.astype(np.float64)))
    err = np.max(np.abs(got.astype(np.float64) - ref.astype(np.float64)) / scale) if ref.size else 0.0
    assert err <= tol, f"{what}: {err}"


def _second(Q, Efull, Z):
    return oracle.double_backward(Q, Efull, Z)


@pytest.mark.parametrize("kind", ["nw", "sw"])
@pytest.mark.parametrize("suffix", ["", "_f32"])
def test_known_answer(golden_dir, kind, suffix):
    """deepblast/tests/test_nw.py:43-54, test_sw.py:42-52 (5x4 fixture, A = 0.1)."""
    d = np.load(os.path.join(golden_dir, f"g2_known_{kind}{suffix}.npz"))
    Vt, E, _, _ = oracle.fwd_bwd(d["theta"], d["A"], None, VAR[kind])
    _close(Vt, d["Vt"], "Vt")
    _close(E, d["E"], "E")
    if suffix == "":
        ref = {"nw": 36.8410610569, "sw": 24.7786304713}[kind]  # SURVEY.md section 4 probe
        assert abs(float(Vt[0]) - ref) < 1e-9


@pytest.mark.parametrize("kind", ["nw", "sw"])
def test_config1_b4_64(golden_dir, kind):
    """BASELINE.json configs[0]: B=4, N=M=64 fp32, incl. double backward and non-uniform Et."""
    d = np.load(os.path.join(golden_dir, f"g1_{kind}_b4_64.npz"))
    for tag, Et in (("", None), ("_et", d["Et"])):
        Vt, E, Q, Efull = oracle.fwd_bwd(d["theta"], d["A"], Et, VAR[kind])
        Ed, Vtd, _ = _second(Q, Efull, d["Z"])
        _close(Vt, d["Vt" + tag], "Vt" + tag)
        _close(E, d["E" + tag], "E" + tag)
        _close(Ed, d["Ed" + tag], "Ed" + tag)
        _close(Vtd, d["Vtd" + tag], "Vtd" + tag)
    assert bool(d["A_grad_is_A"]) and bool(d["A_second_grad_is_None"])


@pytest.mark.parametrize("kind", ["nw", "sw"])
def test_shapes(golden_dir, kind):
    d = np.load(os.path.join(golden_dir, f"g3_{kind}_shapes.npz"))
    for idx in range(len(d["shapes"])):
        p = f"s{idx}_"
        Vt, E, Q, Efull = oracle.fwd_bwd(d[p + "theta"], d[p + "A"], d[p + "Et"], VAR[kind])
        Ed, Vtd, _ = _second(Q, Efull, d[p + "Z"])
        for name, got in (("Vt", Vt), ("E", E), ("Ed", Ed), ("Vtd", Vtd)):
            _close(got, d[p + name], f"{kind} shape {d['shapes'][idx]} {name}")


@pytest.mark.parametrize("kind", ["nw", "sw"])
def test_float64_tensors(golden_dir, kind):
    d = np.load(os.path.join(golden_dir, f"g3_{kind}_f64.npz"))
    Vt, E, Q, Efull = oracle.fwd_bwd(d["theta"], d["A"], None, VAR[kind])
    Ed, Vtd, _ = _second(Q, Efull, d["Z"])
    for name, got in (("Vt", Vt), ("E", E), ("Ed", Ed), ("Vtd", Vtd)):
        _close(got, d[name], name)


@pytest.mark.parametrize("kind", ["nw", "sw"])
def test_lengths_fixture(golden_dir, kind):
    """Per-item sliced calls (deepblast/alignment.py:165-170)."""
    import parity
    d = np.load(os.path.join(golden_dir, f"g6_{kind}_lens.npz"))
    ref = parity.oracle_lens(d["theta"], d["A"], None, None, VAR[kind], d["lens"])
    _close(ref["Vt"], d["Vt"], "Vt")
    _close(ref["E"], d["E"], "E")


@pytest.mark.parametrize("name", ["g5_nw_512", "g5_nw_1024", "g5_sw_512"])
def test_large_checksums(golden_dir, name):
    """512^2 / 1024^2: inputs regenerated from datagen seeds, outputs as samples + checksums."""
    d = np.load(os.path.join(golden_dir, name + ".npz"))
    B, N = int(d["B"]), int(d["N"])
    theta, A = datagen.theta_A(int(d["seed"]), B, N, N)
    Vt, E, _, _ = oracle.fwd_bwd(theta, A, None, VAR[name.split("_")[1]], omp=True)
    _close(Vt, d["Vt"], "Vt")
    _close(E[:, ::61, :], d["E_rows"], "E rows")
    _close(np.stack([np.diagonal(e) for e in E]), d["E_diag"], "E diag")
    assert np.allclose(E.astype(np.float64).sum(axis=(1, 2)), d["E_sum"], rtol=1e-9)


def test_omp_build_matches_serial():
    theta, A = datagen.theta_A(11, 5, 33, 29)
    a = oracle.fwd_bwd(theta, A, None, oracle.NW, omp=False)
    b = oracle.fwd_bwd(theta, A, None, oracle.NW, omp=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
