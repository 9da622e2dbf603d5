"""CPU: host-side logic of the drop-in -- traceback, error behaviour, autograd wiring and the
reference's gradient quirks -- with the kernels replaced by the oracle-backed
fake engine (tests/fake_engine.py).  Parity of the real kernels is tests/test_parity_gpu.py."""
import os

import numpy as np
import pytest
import torch

import deepblast_amd
from deepblast_amd import _engine
from deepblast_amd import NeedlemanWunschDecoder, SmithWatermanDecoder
from fake_engine import OracleEngine

DEC = {"nw": NeedlemanWunschDecoder, "sw": SmithWatermanDecoder}


@pytest.fixture()
def fake(monkeypatch):
    monkeypatch.setattr(_engine, "_ENGINE", OracleEngine())


def test_traceback_matches_reference_fixtures(golden_dir):
    """deepblast/nw.py:401-444 incl. the walks that leave the matrix (IndexError)."""
    d = np.load(os.path.join(golden_dir, "g8_tracebacks.npz"))
    dec = NeedlemanWunschDecoder("softmax")
    n_ok = 0
    for k in range(int(d["count"])):
        g = torch.from_numpy(d[f"t{k}_grad"])
        if bool(d[f"t{k}_ok"]):
            assert np.array_equal(np.array(dec.traceback(g)), d[f"t{k}_states"]), k
            n_ok += 1
        else:
            with pytest.raises(IndexError):
                dec.traceback(g)
    assert n_ok >= 15


def test_cuda_rule_traceback_matches_the_gpu_classes_fixtures(golden_dir):
    """traceback_rule="cuda": the walk of the classes this package replaces (deepblast/nw_cuda.py:273-317,
    sw_cuda.py:283-327 -- stop when ANY neighbour is off the matrix, sentinel -1e10).  Fixtures: what the real
    classes' traceback returned (oracle/gen_golden_tb_cuda.py)."""
    d = np.load(os.path.join(golden_dir, "g12_tracebacks_cuda.npz"))
    n_diff = 0
    for kind in ("nw", "sw"):
        dec, cpu = DEC[kind]("softmax", traceback_rule="cuda"), DEC[kind]("softmax")
        assert cpu.traceback_rule == "cpu"
        for k in range(int(d["count"])):
            g = torch.from_numpy(d[f"t{k}_grad"])
            got = dec.traceback(g)
            assert np.array_equal(np.array(got).reshape(-1, 3), d[f"t{k}_{kind}"]), (kind, k)
            try:
                n_diff += got != cpu.traceback(g)
            except IndexError:
                n_diff += 1   # the CPU rule walks off this matrix; the cuda rule never does
    assert n_diff > 0   # the fixtures do exercise the difference between the two rules
    with pytest.raises(ValueError):
        NeedlemanWunschDecoder("softmax", traceback_rule="gpu")


@pytest.mark.parametrize("kind", ["nw", "sw"])
def test_known_answer_traceback(golden_dir, fake, kind):
    """test_nw.py:43-54 / test_sw.py:42-52: decode the 5x4 fixture end to end (fp32 path)."""
    d = np.load(os.path.join(golden_dir, f"g2_known_{kind}_f32.npz"))
    theta = torch.from_numpy(d["theta"]).requires_grad_()
    A = torch.from_numpy(d["A"]).requires_grad_()
    dec = DEC[kind]("softmax")
    v = dec(theta, A)
    v.backward()
    assert dec.traceback(theta.grad.squeeze()) == [tuple(r) for r in d["traceback"].tolist()]
    want = {"nw": [(0, 0, 0), (1, 0, 0), (2, 0, 1), (3, 1, 1), (4, 2, 2), (4, 3, 1)],
            "sw": [(-1, 0, 1), (0, 1, 0), (1, 1, 0), (2, 1, 0), (3, 1, 1), (4, 2, 2), (4, 3, 1)]}[kind]
    assert dec.traceback(theta.grad.squeeze()) == want


def test_error_behaviour_matches_gpu_variant(fake):
    """nw_cuda.py:171-175: NotImplementedError for other operators, TypeError for dtypes the sweeps do not have (float64 is
    taken, like the reference's CPU classes take it: tests/test_float64_gpu.py) and for theta / A of two dtypes."""
    th = torch.rand(1, 4, 4)
    A = -torch.rand(1, 4, 4)
    with pytest.raises(NotImplementedError):
        NeedlemanWunschDecoder("sparsemax")(th, A)
    with pytest.raises(NotImplementedError):
        NeedlemanWunschDecoder(None)(th, A)
    with pytest.raises(TypeError):
        NeedlemanWunschDecoder("softmax")(th.half(), A.half())
    with pytest.raises(TypeError):
        NeedlemanWunschDecoder("softmax")(th.double(), A)
    with pytest.raises(ValueError):
        NeedlemanWunschDecoder("softmax")(th, A[:, :3])
    SmithWatermanDecoder(None)(th, A)        # CPU reference is built with operator=None (test_sw.py:40)
    SmithWatermanDecoder("softmax")(th, A)   # GPU reference with 'softmax' (alignment.py:74)


def test_real_engine_refuses_cpu_tensors():
    """No silent CPU fallback: the product engine fails loudly off-GPU."""
    th = torch.rand(1, 4, 4)
    A = -torch.rand(1, 4, 4)
    with pytest.raises((RuntimeError, ImportError)):
        NeedlemanWunschDecoder("softmax")(th, A)


@pytest.mark.parametrize("kind", ["nw", "sw"])
def test_autograd_wiring_and_quirks(golden_dir, fake, kind):
    d = np.load(os.path.join(golden_dir, f"g1_{kind}_b4_64.npz"))
    dec = DEC[kind]("softmax")
    theta = torch.from_numpy(d["theta"]).requires_grad_()
    A = torch.from_numpy(d["A"]).requires_grad_()
    Vt = dec(theta, A)
    assert np.array_equal(Vt.detach().numpy(), d["Vt"])
    Vt.backward(torch.from_numpy(d["Et"]))
    assert np.array_equal(theta.grad.numpy(), d["E_et"])
    assert torch.equal(A.grad, A.detach())  # quirk 1: first-order "grad" of A is A (nw.py:337-339,355)

    # decode() -> weighted sum -> backward: Ed in theta.grad, nothing for A (quirk 2, nw.py:386)
    theta.grad = None
    A.grad = None
    aln = dec.decode(theta, A)
    assert aln.shape == theta.shape and aln.requires_grad
    (aln * torch.from_numpy(d["Z"])).sum().backward()
    assert np.array_equal(theta.grad.numpy(), d["Ed"])
    assert A.grad is None

    # Vtd: gradient w.r.t. Et of <E, Z> (nw.py:386 third slot)
    et = torch.from_numpy(d["Et"]).requires_grad_()
    Vt2 = dec(theta, A)
    g, _ = torch.autograd.grad(Vt2, (theta, A), et, create_graph=True)
    (vtd,) = torch.autograd.grad((g * torch.from_numpy(d["Z"])).sum(), et)
    assert np.allclose(vtd.numpy(), d["Vtd_et"], rtol=0, atol=0)


def test_lengths_extension(golden_dir, fake):
    d = np.load(os.path.join(golden_dir, "g6_nw_lens.npz"))
    dec = NeedlemanWunschDecoder("softmax")
    theta = torch.from_numpy(d["theta"]).requires_grad_()
    A = torch.from_numpy(d["A"])
    Vt = dec(theta, A, torch.from_numpy(d["lens"]))
    Vt.sum().backward()
    assert np.array_equal(Vt.detach().numpy(), d["Vt"])
    assert np.array_equal(theta.grad.numpy(), d["E"])


def test_public_surface():
    for name in ("NeedlemanWunschDecoder", "NeedlemanWunschFunction", "NeedlemanWunschFunctionBackward",
                 "SmithWatermanDecoder", "SmithWatermanFunction", "SmithWatermanFunctionBackward"):
        assert hasattr(deepblast_amd, name)
    assert deepblast_amd.NeedlemanWunschFunction.__name__ == "NeedlemanWunschFunction"
    dec = NeedlemanWunschDecoder("softmax")
    assert dec.operator == "softmax" and isinstance(dec, torch.nn.Module)


def test_wide_problems_are_swept_transposed(fake):
    """More columns than sdp_max_cols(): the decoder sweeps the transposed problem (the recurrence is symmetric in its axes)
    and autograd transposes the results back -- here through the oracle-backed engine: values and quirks as for any shape."""
    from oracle import oracle
    B, N, M = 2, 5, 2100
    rng = np.random.default_rng(5)
    theta = rng.random((B, N, M), dtype=np.float32)
    A = -rng.random((B, N, M), dtype=np.float32)
    Z = rng.standard_normal((B, N, M)).astype(np.float32)
    for name, variant in (("nw", 0), ("sw", 1)):
        dec = DEC[name]("softmax")
        t = torch.from_numpy(theta).requires_grad_()
        a = torch.from_numpy(A).requires_grad_()
        aln = dec.decode(t, a)
        assert aln.shape == (B, N, M)
        (aln * torch.from_numpy(Z)).sum().backward()
        Vt, E, Q, Ef = oracle.fwd_bwd(theta, A, None, variant)
        Ed, _, _ = oracle.double_backward(Q, Ef, Z)
        assert np.abs(aln.detach().numpy() - E).max() <= 1e-6 and np.abs(t.grad.numpy() - Ed).max() <= 1e-5 * max(1.0, np.abs(Ed).max())
        assert a.grad is None
        t2 = torch.from_numpy(theta).requires_grad_()
        a2 = torch.from_numpy(A).requires_grad_()
        v = dec(t2, a2)
        assert np.abs(v.detach().numpy() - Vt).max() <= 1e-4
        v.sum().backward()
        assert np.abs(t2.grad.numpy() - E).max() <= 1e-6 and torch.equal(a2.grad, a2.detach())
