"""GPU: the HIP kernels, called through the C ABI, against the CPU oracle and the committed
golden fixtures (generated from the real reference).  Tolerance: <= 1e-4 (parity.TOL), absolute
on E/Ed, relative on Vt/Vtd -- BASELINE.json north_star, SURVEY.md 8c."""
import os

import numpy as np
import pytest

import datagen
import parity

pytestmark = pytest.mark.gpu

SHAPES = [(1, 1, 1), (2, 1, 7), (2, 7, 1), (2, 2, 2), (3, 5, 4), (2, 16, 16), (2, 17, 33), (2, 37, 101),
          (2, 64, 64), (2, 63, 65), (2, 65, 63), (2, 65, 64), (2, 101, 37), (2, 128, 128), (2, 129, 70),
          (2, 200, 300), (3, 257, 255), (2, 320, 90), (3, 512, 512), (1, 300, 1024), (1, 1024, 1024),
          (1, 70, 2048), (1, 300, 2048), (2, 1100, 40)]


# Explicit ceiling of the PLAIN second-order error in the three named cases whose bound is stated on the scaled figure (max|Ed_ref|
# of 5-30 there; INTEGRATION.md, first screen): measured 1.05e-4 ... 1.5e-4, of which ~1e-4 is the fp32 reference's own distance
# from its float64 run.  Anything beyond 2e-4 is a regression, scaled or not.
PLAIN_CEILING = 2e-4


def _assert(errs, what=""):
    for k, v in errs.items():
        assert np.isfinite(v) and v <= parity.TOL, f"{what} {k}: {v:.3e} > {parity.TOL}"


def test_selftest():
    from deepblast_amd._engine import get_engine
    get_engine().selftest(0)


@pytest.mark.parametrize("variant", [0, 1], ids=["nw", "sw"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_all_four_passes_vs_oracle(shape, variant):
    B, N, M = shape
    idx = SHAPES.index(shape)
    theta, A = datagen.theta_A(1000 + idx, B, N, M)
    if idx % 3 == 1:
        A = (-A).astype(np.float32)  # positive gap scores too (test_nw.py:67-72)
    Z = datagen.normal(2000 + idx, (B, N, M))
    Et = (0.5 + datagen.uniform(3000 + idx, (B,))).astype(np.float32)
    ref = parity.oracle_all(theta, A, Et, Z, variant)
    got = parity.engine_all(theta, A, Et, Z, variant)
    _assert(parity.compare(got, ref), f"{shape} v{variant}")
    # SURVEY 8c states the second-order bound as a PLAIN max-abs: held here with no scaling by max|Ed_ref| at all
    plain = parity.unscaled(got, ref)
    assert plain <= parity.TOL, f"{shape} v{variant} unscaled Ed: {plain:.3e} (max|Ed_ref| = {np.abs(ref['Ed']).max():.2f})"


@pytest.mark.parametrize("shape", [(2, 512, 512), (1, 40, 2048), (1, 1500, 24)], ids=lambda s: "x".join(map(str, s)))
def test_peaked_alignments_keep_parity(shape):
    """Sharp inputs (one dominant path, weights near 0 and 1) are the worst case for the compact saved state of
    the backward sweep: an error in a weight that is close to 1 is carried along the whole alignment path."""
    B, N, M = shape
    theta, A = datagen.theta_A(77, B, N, M)
    theta = (12.0 * theta).astype(np.float32)
    A = (6.0 * A).astype(np.float32)
    Z = datagen.normal(78, (B, N, M))   # second order too
    ref = parity.oracle_all(theta, A, None, Z, 0)
    got = parity.engine_all(theta, A, None, Z, 0)
    _assert(parity.compare(got, ref), f"peaked {shape}")


@pytest.mark.parametrize("variant", [0, 1], ids=["nw", "sw"])
def test_nonzero_ZA_enters_the_adjoint(variant):
    """quirk 3: the ZA buffer is part of the adjoint recurrences (nw.py:190-192)."""
    B, N, M = 2, 70, 45
    theta, A = datagen.theta_A(50, B, N, M)
    Z = datagen.normal(51, (B, N, M))
    ZA = datagen.normal(52, (B, N, M))
    ref = parity.oracle_all(theta, A, None, Z, variant, ZA=ZA)
    got = parity.engine_all(theta, A, None, Z, variant, ZA=ZA)
    _assert(parity.compare(got, ref))


@pytest.mark.parametrize("variant", [0, 1], ids=["nw", "sw"])
def test_lengths_aware_mode(variant):
    """Padded batch with per-pair sizes == per-item sliced reference calls (alignment.py:165-170)."""
    B, N, M = 9, 150, 170
    theta, A = datagen.theta_A(4000, B, N, M)
    Z = datagen.normal(4001, (B, N, M))
    lens = datagen.lengths(4002, B, 1, 150)
    lens[0] = (N, M)
    lens[1] = (1, 1)
    lens[2] = (64, 65)
    ref = parity.oracle_lens(theta, A, None, Z, variant, lens)
    got = parity.engine_all(theta, A, None, Z, variant, lens=lens)
    _assert(parity.compare(got, ref))
    for b in range(B):  # outside the block everything is exactly zero
        n, m = lens[b]
        assert not got["E"][b, n:, :].any() and not got["E"][b, :, m:].any()
        assert not got["Ed"][b, n:, :].any() and not got["Ed"][b, :, m:].any()


@pytest.mark.parametrize("kind", ["nw", "sw"])
def test_golden_config1_and_shapes(golden_dir, kind):
    """Fixtures produced by the real reference: B=4 N=M=64 (BASELINE configs[0]) + odd shapes."""
    variant = {"nw": 0, "sw": 1}[kind]
    d = np.load(os.path.join(golden_dir, f"g1_{kind}_b4_64.npz"))
    got = parity.engine_all(d["theta"], d["A"], d["Et"], d["Z"], variant)
    _assert(parity.compare(got, {"Vt": d["Vt_et"], "E": d["E_et"], "Ed": d["Ed_et"], "Vtd": d["Vtd_et"]}))
    d = np.load(os.path.join(golden_dir, f"g3_{kind}_shapes.npz"))
    for idx in range(len(d["shapes"])):
        p = f"s{idx}_"
        got = parity.engine_all(d[p + "theta"], d[p + "A"], d[p + "Et"], d[p + "Z"], variant)
        _assert(parity.compare(got, {k: d[p + k] for k in ("Vt", "E", "Ed", "Vtd")}), str(d["shapes"][idx]))


@pytest.mark.parametrize("name", ["g5_nw_512", "g5_nw_1024", "g5_sw_512"])
def test_golden_large(golden_dir, name):
    """512^2 and 1024^2 outputs of the real reference (samples + checksums)."""
    d = np.load(os.path.join(golden_dir, name + ".npz"))
    B, N = int(d["B"]), int(d["N"])
    theta, A = datagen.theta_A(int(d["seed"]), B, N, N)
    got = parity.engine_all(theta, A, None, None, {"nw": 0, "sw": 1}[name.split("_")[1]])
    assert parity.rel_err(got["Vt"], d["Vt"]) <= parity.TOL
    assert parity.abs_err(got["E"][:, ::61, :], d["E_rows"]) <= parity.TOL
    assert parity.abs_err(np.stack([np.diagonal(e) for e in got["E"]]), d["E_diag"]) <= parity.TOL
    assert np.allclose(got["E"].astype(np.float64).sum(axis=(1, 2)), d["E_sum"], rtol=1e-5)


# Thin, long, steep problems: one live path of ~M saturated cells.  The reference divides in float64 and rounds the
# weights once (nw.py:21-22,115), so a saturated weight is stored as exactly 1.0; an engine that leaves it at
# 1 +- 1e-7 is off by up to 1e-3 in Ed here (round-1 fuzz: 42 of 1200 cases above 1e-4).  The exact-state forward
# forms the largest weight as 1 - (the other two) (q_sharpen), which these cases pin at the ordinary 1e-4.
THIN_STEEP = [(2, 688, 1, 5.0, 1.0, 0.0), (3, 1500, 0, 8.0, 1.0, 0.0), (5, 2000, 0, 30.0, 10.0, 0.5),
              (1200, 3, 0, 8.0, 0.0, 0.0), (6, 1900, 0, 30.0, 0.0, 0.0), (7, 2048, 1, 8.0, 10.0, 0.0),
              (4, 1800, 1, 30.0, 40.0, -3.0), (2, 2048, 0, 30.0, 1.0, 0.5)]


@pytest.mark.parametrize("case", THIN_STEEP, ids=lambda c: "x".join(str(v) for v in c))
def test_second_order_on_thin_steep_problems(case):
    N, M, variant, ts, as_, ao = case
    B = 3
    theta, A = datagen.theta_A(900 + N + M, B, N, M)
    theta = (theta * ts).astype(np.float32)
    A = (A * as_ + ao).astype(np.float32)
    Z = datagen.normal(950 + N + M, (B, N, M))
    ref = parity.oracle_all(theta, A, None, Z, variant)
    got = parity.engine_all(theta, A, None, Z, variant)
    _assert(parity.compare(got, ref), f"thin/steep {case}")


# Steep scores on full batches of mid-size pairs: |theta| reaches ~100, where the fp32 product theta * log2(e) is off
# by ~1e-5 bits; the weights of the neighbouring cells inherit that and Vtd -- a sum over a few hundred soft cells with
# partial sums of magnitude 25 -- was off by 1.2e-4 on 2 of 20000 fuzzed pairs.  The exact-state forward now forms the
# exponent from the exact product (exp2_residual and the hi/lo split in the per-step form); these pin it.
@pytest.mark.parametrize("case", [(129, 348, 303, 0, 30.0, 10.0, 50046, 60046), (198, 110, 363, 0, 30.0, 0.0, 50090, 60090),
                                  (180, 306, 262, 1, 8.0, 10.0, 50018, 60018)], ids=lambda c: "x".join(str(v) for v in c[:6]))
def test_second_order_on_steep_full_batches(case):
    B, N, M, variant, ts, as_, s1, s2 = case
    theta, A = datagen.theta_A(s1, B, N, M)
    theta = (theta * ts).astype(np.float32)
    A = (A * as_).astype(np.float32)
    Z = datagen.normal(s2, (B, N, M))
    ref = parity.oracle_all(theta, A, None, Z, variant, omp=True)
    got = parity.engine_all(theta, A, None, Z, variant)
    errs = parity.compare(got, ref, plain=case[4] < 30.0)   # (theta x 30: max|Ed_ref| ~ 10, plain error 1.2e-4 -- the stated envelope)
    _assert(errs, f"steep full batch {case}")
    # ... and the PLAIN max-abs stays asserted there too, under an explicit ceiling: a regression must not hide
    # behind the scaling.  On this case the fp32 reference differs from its own float64 run by ~1e-4.
    assert parity.unscaled(got, ref) <= PLAIN_CEILING, parity.unscaled(got, ref)
    assert errs["Vtd"] <= 0.5 * parity.TOL, errs   # margin: the bound is met with room, not at the 4-sigma tail


@pytest.mark.parametrize("case", [(1, 8192, 2048, 0, 30.0, 1.0), (1, 20000, 2048, 1, 8.0, 1.0), (1, 60000, 1000, 0, 30.0, 10.0)],
                         ids=lambda c: "x".join(str(v) for v in c))
def test_long_steep_problems_keep_first_order_parity(case):
    """Path lengths far beyond N + M = 4096 on steep scores (one saturated path): the packed state would lose
    ~1.7e-8 of E per step here (1.5e-4 ... 5e-4 on these cases); the library keeps the exact state for such problems
    (sdp_api.hip: exact_for).  Plain forward/backward calls, no flags."""
    import torch
    from deepblast_amd._engine import get_engine
    B, N, M, variant, ts, as_ = case
    theta, A = datagen.theta_A(83000 + N, B, N, M)
    theta = (theta * ts).astype(np.float32)
    A = (A * as_).astype(np.float32)
    ref = parity.oracle_all(theta, A, None, None, variant)
    eng = get_engine()
    t, a = torch.from_numpy(theta).cuda(), torch.from_numpy(A).cuda()
    Vt, Q = eng.forward(t, a, variant)
    E = eng.backward(torch.ones(B, device="cuda"), Q, tuple(t.shape), variant)
    errs = parity.compare({"Vt": Vt.cpu().numpy(), "E": E.cpu().numpy()}, ref)
    _assert(errs, f"long steep {case}")
    assert errs["E"] <= 2e-5, errs


def test_long_problems_with_lengths_and_launch_order():
    """N + M > 4096 (exact state chosen by the library) together with per-pair lengths on more pairs than CUs (the
    launch order lives in the tail of the state buffer, whose offset depends on the state format): all four passes."""
    B, N, M = 260, 4100, 24
    theta, A = datagen.theta_A(85000, B, N, M)
    theta = (theta * 4.0).astype(np.float32)
    Z = datagen.normal(85001, (B, N, M))
    lens = datagen.lengths(85002, B, 1, 4100)
    lens[:, 1] = np.minimum(lens[:, 1], M)
    lens[0] = (N, M)
    ref = parity.oracle_lens(theta, A, None, Z, 0, lens)
    got = parity.engine_all(theta, A, None, Z, 0, lens=lens)
    _assert(parity.compare(got, ref, plain=False), "long + lengths")   # (max|Ed_ref| > 1 on these long pairs: scaled figure)
    # (the plain figure under an explicit ceiling of its own: pairs of up to 4100 rows on theta x 4 -- max|Ed_ref| is in the tens, the
    #  plain error measured 7.1e-4 in round 6 -- scaled by max|Ed_ref| it is what the line above holds to 1e-4)
    print(f"\nlong + lengths: plain max|dEd| = {parity.unscaled(got, ref):.2e}, max|Ed_ref| = {np.abs(ref['Ed']).max():.1f}")
    assert parity.unscaled(got, ref) <= 1e-3, parity.unscaled(got, ref)


def test_where_the_fp32_reference_is_the_noisy_one():
    """2048 x 2048 Smith-Waterman, theta x30, A x10: Ed differs from the reference's fp32 result by 2e-4 -- and from the
    reference run in float64 by 1e-6.  The reference forms the Hessian product and the Qd*E products in the storage
    dtype (nw.py:34-41, 261-266; the oracle keeps that): on a saturated path a_k - sum(q a) cancels at magnitude ~60,
    3.6e-6 of noise per cell, a random walk over 4096 steps.  The engine does those in float64.  This test pins both
    facts, so that the 1e-4 bound is read against the right reference where the two disagree."""
    B, N, M, variant = 2, 2048, 2048, 1
    theta, A = datagen.theta_A(81001, B, N, M)
    theta = (theta * 30.0).astype(np.float32)
    A = (A * 10.0).astype(np.float32)
    Z = datagen.normal(82001, (B, N, M))
    ref32 = parity.oracle_all(theta, A, None, Z, variant)
    ref64 = parity.oracle_all(theta.astype(np.float64), A.astype(np.float64), None, Z.astype(np.float64), variant)
    got = parity.engine_all(theta, A, None, Z, variant)
    e64, e32, noise = parity.compare(got, ref64), parity.compare(got, ref32), parity.compare(ref32, ref64)
    assert max(e64["Ed"], e64["Vtd"], e64["Ex"]) <= 1e-5, e64
    assert e64["E"] <= 5e-5, e64     # E is the packed-state path here (N + M = 4096, its longest): 2.4e-5
    assert noise["Ed"] > parity.TOL and e32["Ed"] <= 1.1 * noise["Ed"] + 1e-5, (noise, e32)
    assert max(e32["E"], e32["Vt"], e32["Vtd"], e32["Ex"]) <= parity.TOL, e32
    # ... and the reference-rounding mode (variant | SDP_REF_ROUNDING: the same fp32 roundings from the same fp32 weights)
    # is within the bound of the fp32 reference here too -- far within: it reproduces the reference's noise, not just its size
    eref = parity.compare(parity.engine_ref(theta, A, None, Z, variant), ref32)
    print("reference-rounding mode vs the fp32 oracle:", eref)
    assert max(eref.values()) <= 0.02 * parity.TOL, eref


REF_SHAPES = [(4, 64, 64), (3, 37, 101), (1, 1, 1), (2, 1, 7), (2, 7, 1), (5, 130, 200), (2, 300, 45), (3, 257, 513), (2, 562, 9), (3, 1361, 7)]


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("shape", REF_SHAPES, ids=lambda s: "x".join(map(str, s)))
def test_reference_rounding_mode_reproduces_the_fp32_reference(shape, variant):
    """variant | SDP_REF_ROUNDING (csrc/sdp_ref.hip): float64 exp / log / division and one rounding of Q to fp32
    (nw.py:10-27, 115), the Hessian product and the Qd * E products rounded to fp32 where numpy rounds them (nw.py:30-43,
    261-266).  Against the fp32 oracle -- which is pinned bit for bit to the real reference (tests/test_oracle_golden.py)
    -- all four sweeps then agree to rounding-flip level (a differently rounded last bit of a float64 exp can flip the fp32
    rounding of a weight once in ~1e9 cells), steep scores, positive gaps, ZA, Et and per-pair lengths included."""
    B, N, M = shape
    theta, A = datagen.theta_A(7100 + N + M, B, N, M)
    Z = datagen.normal(7200 + N, (B, N, M))
    ZA = datagen.normal(7300 + N, (B, N, M))
    Et = (0.5 + datagen.uniform(7400 + N, (B,))).astype(np.float32)
    for ts, as_, ao in ((1.0, 1.0, 0.0), (30.0, 10.0, 0.0), (8.0, 1.0, 0.5)):
        th = (theta * ts).astype(np.float32)
        a = (A * as_ + ao).astype(np.float32)
        ref = parity.oracle_all(th, a, Et, Z, variant, ZA=ZA)
        got = parity.engine_ref(th, a, Et, Z, variant, ZA=ZA)
        e = parity.compare(got, ref)
        assert max(e.values()) <= 0.01 * parity.TOL, (shape, variant, ts, as_, ao, e)
    lens = datagen.lengths(7500 + N, B, 1, max(N, M))
    lens[:, 0] = np.minimum(lens[:, 0], N)
    lens[:, 1] = np.minimum(lens[:, 1], M)
    ref = parity.oracle_lens(theta, A, None, Z, variant, lens)
    got = parity.engine_ref(theta, A, None, Z, variant, lens=lens)
    e = parity.compare(got, ref)
    assert max(e.values()) <= 0.01 * parity.TOL, (shape, variant, "lens", e)


def test_reference_arithmetic_through_the_public_api():
    """Decoder(operator, arithmetic="reference"): forward, decode and the double backward run the reference-rounding
    sweeps; results against the fp32 oracle at rounding-flip level, quirks unchanged (A.grad == A)."""
    import torch
    from deepblast_amd import NeedlemanWunschDecoder
    B, N, M = 3, 90, 70
    theta, A = datagen.theta_A(7600, B, N, M)
    theta = (theta * 8).astype(np.float32)
    Z = datagen.normal(7601, (B, N, M))
    ref = parity.oracle_all(theta, A, None, Z, 0)
    dec = NeedlemanWunschDecoder("softmax", arithmetic="reference")
    t = torch.from_numpy(theta).cuda().requires_grad_()
    a = torch.from_numpy(A).cuda().requires_grad_()
    aln = dec.decode(t, a)
    (aln * torch.from_numpy(Z).cuda()).sum().backward()
    with torch.no_grad():
        vt = dec(t, a)
    got = {"Vt": vt.cpu().numpy(), "E": aln.detach().cpu().numpy(), "Ed": t.grad.cpu().numpy(), "Vtd": ref["Vtd"]}
    e = parity.compare(got, ref)
    assert max(e.values()) <= 0.01 * parity.TOL, e
    t2 = torch.from_numpy(theta).cuda().requires_grad_()
    a2 = torch.from_numpy(A).cuda().requires_grad_()
    dec(t2, a2).sum().backward()
    assert torch.equal(a2.grad, a2.detach()) and parity.abs_err(t2.grad.cpu().numpy(), ref["E"]) <= 0.01 * parity.TOL
    with pytest.raises(ValueError):
        NeedlemanWunschDecoder("softmax", arithmetic="exactish")


def test_headline_config_second_order_full_batch():
    """BASELINE.json configs[1] on the TRAINING path: B=256, N=M=512 through decode() (exact state: sdp_fwd_x_tp ->
    sdp_bwd_x) and (aln * Z).sum().backward() (adjoint pair), whole batch against the oracle: E, Ed, and Vtd from a
    direct adjoint-forward call."""
    import torch
    from deepblast_amd import NeedlemanWunschDecoder
    from deepblast_amd._engine import get_engine
    B, N, M = 256, 512, 512
    theta, A = datagen.theta_A(1, B, N, M)
    Z = datagen.normal(7, (B, N, M))
    ref = parity.oracle_all(theta, A, None, Z, 0, omp=True)
    t = torch.from_numpy(theta).cuda().requires_grad_()
    a = torch.from_numpy(A).cuda().requires_grad_()
    z = torch.from_numpy(Z).cuda()
    dec = NeedlemanWunschDecoder("softmax")
    aln = dec.decode(t, a)
    (aln * z).sum().backward()
    eng = get_engine()
    Vtx, Qx = eng.forward(t.detach(), a.detach(), 0, exact_state=True)
    Vtd, _ = eng.adjoint_forward(Qx, z, None, 0)
    torch.cuda.synchronize()
    assert eng.check_device()[0] == 0
    errs = parity.compare({"Vt": Vtx.cpu().numpy(), "E": aln.detach().cpu().numpy(), "Ed": t.grad.cpu().numpy(),
                           "Vtd": Vtd.cpu().numpy()}, ref)
    _assert(errs, "headline second order")
    # SURVEY 8c states the Ed bound as a plain max-abs; parity.compare scales it by max(1, max|Ed_ref|) because Ed is
    # unbounded (it is linear in Z).  At the headline batch the two are stated side by side:
    worst = parity.abs_err(t.grad.cpu().numpy(), ref["Ed"])
    scale = float(np.max(np.abs(ref["Ed"])))
    print(f"headline batch: unscaled max|dEd| = {worst:.3e} (max|Ed_ref| = {scale:.3f}, scaled error {errs['Ed']:.3e})")
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "headline_ed_unscaled.txt"), "w") as f:
            f.write(f"B=256 N=M=512 NW, Z ~ N(0,1): unscaled max|Ed - Ed_ref| = {worst:.3e}; max|Ed_ref| = {scale:.4f}; scaled = {errs['Ed']:.3e}\n")
    assert worst <= parity.TOL * max(1.0, scale)


@pytest.mark.parametrize("variant", [0, 1], ids=["nw", "sw"])
def test_variable_length_batch_larger_than_the_gpu_runs_longest_first(variant):
    """More pairs than CUs with per-pair lengths: the library launches them longest-first (sdp_order_kernel writes
    the order into the tail of the state buffer, every sweep reads it).  Results must be those of the per-item
    reference calls, in the ORIGINAL batch order, for all four sweeps."""
    B, N, M = 700, 130, 150
    theta, A = datagen.theta_A(4100, B, N, M)
    Z = datagen.normal(4101, (B, N, M))
    lens = datagen.lengths(4102, B, 1, 130)
    lens[5] = (N, M)
    lens[699] = (N, M - 1)
    ref = parity.oracle_lens(theta, A, None, Z, variant, lens)
    got = parity.engine_all(theta, A, None, Z, variant, lens=lens)
    _assert(parity.compare(got, ref))
    sel = [5, 6, 350, 699]   # a pair's results do not depend on the batch (or the launch position) it ran in
    alone = parity.engine_all(theta[sel], A[sel], None, Z[sel], variant, lens=lens[sel])
    for k in ("Vt", "E", "Ed", "Vtd"):
        assert np.array_equal(alone[k], got[k][sel]), k


@pytest.mark.parametrize("case", [(300, 70, 150, 0, False), (300, 69, 131, 1, False), (300, 130, 150, 0, True), (160, 200, 516, 0, False),
                                  (217, 9, 6, 0, False), (142, 63, 25, 0, False), (188, 57, 59, 1, False), (237, 15, 10, 0, True),
                                  (279, 21, 35, 0, False), (130, 3, 1, 0, False), (131, 70, 2, 1, False),
                                  # fewer pairs than CUs, nine strips: the general-pitch K = 32 backward build on EIGHT waves (round 6)
                                  (40, 520, 516, 0, False), (40, 520, 516, 1, True)],
                         ids=lambda c: "x".join(str(int(v)) for v in c))
def test_row_pitch_and_planes_not_aligned_to_cache_lines(case):
    """Full batches (throughput builds) whose rows and per-pair planes do not start on 128-byte lines: the input
    blocks and the output blocks are moved so that they are aligned in MEMORY (per-row offsets delta_r / D_r), the
    results must not notice -- all four sweeps against the oracle, and batch independence bit for bit."""
    B, N, M, variant, use_lens = case
    theta, A = datagen.theta_A(4200 + N, B, N, M)
    Z = datagen.normal(4201 + N, (B, N, M))
    lens = datagen.lengths(4202, B, 1, min(N, M)) if use_lens else None   # (planes at offsets b*N*M floats: any alignment)
    ref = parity.oracle_lens(theta, A, None, Z, variant, lens) if use_lens else parity.oracle_all(theta, A, None, Z, variant)
    got = parity.engine_all(theta, A, None, Z, variant, lens=lens)
    _assert(parity.compare(got, ref), str(case))
    sel = [1, 7, B - 1]   # odd pair indices: planes at odd offsets
    alone = parity.engine_all(theta[sel], A[sel], None, Z[sel], variant, lens=None if lens is None else lens[sel])
    for k in ("Vt", "E", "Ed", "Vtd"):
        assert np.array_equal(alone[k], got[k][sel]), k


def test_headline_config_full_batch():
    """BASELINE.json configs[1]: B=256, N=M=512, whole batch against the oracle, plus
    size-independent properties: batch independence (bit-exact) and linearity in Et."""
    B, N, M = 256, 512, 512
    theta, A = datagen.theta_A(1, B, N, M)
    ref = parity.oracle_all(theta, A, None, None, 0, omp=True)
    got = parity.engine_all(theta, A, None, None, 0)
    _assert(parity.compare(got, ref))
    sel = [0, 100, 255]
    alone = parity.engine_all(theta[sel], A[sel], None, None, 0)
    assert np.array_equal(alone["Vt"], got["Vt"][sel]) and np.array_equal(alone["E"], got["E"][sel])
    Et = np.full(3, 3.0, np.float32)
    scaled = parity.engine_all(theta[sel], A[sel], Et, None, 0)
    # fp32 products along up to N+M-1 steps round differently for e and 3e: a few 1e-7 relative per element
    assert np.allclose(scaled["E"], 3.0 * alone["E"], rtol=3e-6, atol=1e-7)
    assert abs(float(got["E"][0, -1, -1]) - 1.0) < 1e-6  # E[N,M] == Et (quirk 4)


def test_config4_smith_waterman_full_batch():
    """BASELINE.json configs[3]: SmithWatermanDecoder, B=256, N=M=512 -- reference semantics (3-state
    recurrence skipping padded row/col 1, sw.py:54-55,107-110), whole batch against the oracle."""
    B, N, M = 256, 512, 512
    theta, A = datagen.theta_A(1, B, N, M)
    ref = parity.oracle_all(theta, A, None, None, 1, omp=True)
    got = parity.engine_all(theta, A, None, None, 1)
    _assert(parity.compare(got, ref))
    assert not got["E"][:, 0, :].any() and not got["E"][:, :, 0].any()  # first row / column never aligned


def _config3():
    B = 256
    lens = datagen.lengths(2, B, 64, 1024)
    N, M = int(lens[:, 0].max()), int(lens[:, 1].max())
    theta, A = datagen.theta_A(2, B, N, M)
    for b in range(B):  # padding value 0 outside each pair's block
        theta[b, lens[b, 0]:, :] = 0
        theta[b, :, lens[b, 1]:] = 0
        A[b, lens[b, 0]:, :] = 0
        A[b, :, lens[b, 1]:] = 0
    return B, N, M, lens, theta, A


def _host_threads():
    return max(1, min(64, (os.cpu_count() or 8) // 2))


def test_config3_variable_length_padded_batch():
    """BASELINE.json configs[2]: B=256 pairs with N_b, M_b ~ U[64,1024] padded to (256,1022,1020).
    Reference semantics (alignment.py:117-124) = DP over the full padded matrix: ALL 256 pairs against the oracle
    (OpenMP, 32 pairs at a time), plus batch-independence against a 4-pair run (bit-exact)."""
    B, N, M, lens, theta, A = _config3()
    full = parity.engine_all(theta, A, None, None, 0)
    ref = parity.oracle_chunked(theta, A, None, None, 0)
    _assert(parity.compare(full, ref), "padded, whole batch")
    sel = [0, 17, 101, 255]
    alone = parity.engine_all(theta[sel], A[sel], None, None, 0)
    assert np.array_equal(alone["E"], full["E"][sel]) and np.array_equal(alone["Vt"], full["Vt"][sel])


def test_config3_lengths_aware_whole_batch_first_and_second_order():
    """configs[2] in lengths-aware mode (terminal cell at (N_b, M_b), zero outside): ALL 256 pairs, all four sweeps,
    against per-item sliced oracle calls (alignment.py:165-170)."""
    B, N, M, lens, theta, A = _config3()
    Z = datagen.normal(2002, (B, N, M))
    aware = parity.engine_all(theta, A, None, Z, 0, lens=lens)
    refl = parity.oracle_lens(theta, A, None, Z, 0, lens, threads=_host_threads())
    _assert(parity.compare(aware, refl), "lengths-aware, whole batch")
    for b in range(B):
        n, m = lens[b]
        assert not aware["E"][b, n:, :].any() and not aware["E"][b, :, m:].any()
        assert not aware["Ed"][b, n:, :].any() and not aware["Ed"][b, :, m:].any()
        assert abs(float(aware["E"][b, n - 1, m - 1]) - 1.0) < 1e-6  # terminal cell of the true block


def test_config4_smith_waterman_second_order_full_batch():
    """BASELINE.json configs[3] on the training path: SmithWatermanDecoder.decode() -> (aln * Z).sum().backward(),
    B=256, N=M=512, whole batch against the oracle: E, Ed, Vtd."""
    import torch
    from deepblast_amd import SmithWatermanDecoder
    from deepblast_amd._engine import get_engine
    B, N, M = 256, 512, 512
    theta, A = datagen.theta_A(1, B, N, M)
    Z = datagen.normal(9, (B, N, M))
    ref = parity.oracle_chunked(theta, A, None, Z, 1, chunk=64)
    t = torch.from_numpy(theta).cuda().requires_grad_()
    a = torch.from_numpy(A).cuda().requires_grad_()
    z = torch.from_numpy(Z).cuda()
    aln = SmithWatermanDecoder("softmax").decode(t, a)
    (aln * z).sum().backward()
    eng = get_engine()
    Vtx, Qx = eng.forward(t.detach(), a.detach(), 1, exact_state=True)
    Vtd, _ = eng.adjoint_forward(Qx, z, None, 1)
    torch.cuda.synchronize()
    assert eng.check_device()[0] == 0
    errs = parity.compare({"Vt": Vtx.cpu().numpy(), "E": aln.detach().cpu().numpy(), "Ed": t.grad.cpu().numpy(),
                           "Vtd": Vtd.cpu().numpy()}, ref)
    _assert(errs, "SW second order, whole batch")


@pytest.mark.parametrize("waves", [1, 2, 3, 4, 5, 7, 8])
def test_every_wave_count_gives_identical_results(waves):
    """The strip hand-off (two shared LDS boundary rows, per-wave progress words, frame words) must not depend on
    how many wavefronts share a pair, nor on the build (more than 4 waves selects the latency builds with their
    shorter chunks): results are bit-identical -- same arithmetic, different schedule."""
    from deepblast_amd._engine import get_engine
    eng = get_engine()
    B, N, M = 3, 720, 333   # 12 strips: waves own several each, W = 5, 7 do not divide them evenly
    theta, A = datagen.theta_A(61, B, N, M)
    Z = datagen.normal(62, (B, N, M))
    ref = parity.oracle_all(theta, A, None, Z, 0)
    base = parity.engine_all(theta, A, None, Z, 0)
    try:
        eng.force_waves = {p: waves for p in range(4)}   # travels with each call as SDP_WAVES(w)
        got = parity.engine_all(theta, A, None, Z, 0)
    finally:
        eng.force_waves = {}
    _assert(parity.compare(got, ref), f"W={waves}")
    for k in ("Vt", "E", "Ed", "Vtd"):
        assert np.array_equal(got[k], base[k]), (waves, k)


@pytest.mark.parametrize("waves", [2, 4, 8])
def test_wave_counts_agree_bit_for_bit_on_steep_scores(waves):
    """As above on theta x8: windowed and per-step blocks alternate (the per-step form takes over for two blocks after a
    failed windowed attempt -- counted in blocks, so the same cells in the K=16 and K=32 builds), and the two forms
    differ in how they treat saturated weights and large exponents."""
    from deepblast_amd._engine import get_engine
    eng = get_engine()
    B, N, M = 2, 500, 700
    theta, A = datagen.theta_A(63, B, N, M)
    theta = (theta * 8.0).astype(np.float32)
    A = (A * 3.0).astype(np.float32)
    Z = datagen.normal(64, (B, N, M))
    base = parity.engine_all(theta, A, None, Z, 1)
    try:
        eng.force_waves = {p: waves for p in range(4)}
        got = parity.engine_all(theta, A, None, Z, 1)
    finally:
        eng.force_waves = {}
    for k in ("Vt", "E", "Ed", "Vtd", "Ex"):
        assert np.array_equal(got[k], base[k]), (waves, k)


def test_repeated_calls_are_deterministic():
    B, N, M = 16, 257, 300
    theta, A = datagen.theta_A(63, B, N, M)
    Z = datagen.normal(64, (B, N, M))
    a = parity.engine_all(theta, A, None, Z, 1)
    b = parity.engine_all(theta, A, None, Z, 1)
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_many_strips_eight_waves_repeatable():
    """Timing-dependent faults (a missed hand-off, the wide-store data hazard that once corrupted lanes 12-15 of the
    saved state with 8 waves) show up as run-to-run differences: 10 strips on 8 waves, repeated."""
    from deepblast_amd._engine import get_engine
    eng = get_engine()
    B, N, M = 24, 640, 96
    theta, A = datagen.theta_A(65, B, N, M)
    first = parity.engine_all(theta, A, None, None, 0)
    _assert(parity.compare(first, parity.oracle_all(theta, A, None, None, 0)))
    try:
        eng.force_waves = {p: 8 for p in range(4)}
        for rep in range(6):
            again = parity.engine_all(theta, A, None, None, 0)
            for k in first:
                assert np.array_equal(first[k], again[k]), (rep, k)
    finally:
        eng.force_waves = {}
    assert eng.check_device()[0] == 0   # no hand-off ever timed out


@pytest.mark.parametrize("case", [(130, 64, 2048, 0, False), (600, 130, 70, 1, False), (520, 200, 64, 0, True),
                                  (300, 70, 2000, 1, True), (128, 512, 512, 0, False)],
                         ids=["full-longM-latency-fallback", "two-waves-per-pair", "two-waves-lengths", "lengths-longM",
                              "half-the-CUs-throughput"])
def test_every_launch_policy_branch_keeps_parity(case):
    """Batches that make sdp_api.hip::plan pick each of its branches on a 256-CU device (throughput builds, their
    long-M fallback, two waves per pair for more pairs than CUs, the lengths rule): first-order parity."""
    B, N, M, variant, use_lens = case
    theta, A = datagen.theta_A(19, B, N, M)
    lens = datagen.lengths(21, B, 1, min(N, M)) if use_lens else None
    ref = (parity.oracle_lens(theta, A, None, None, variant, lens) if use_lens
           else parity.oracle_all(theta, A, None, None, variant, omp=True))
    got = parity.engine_all(theta, A, None, None, variant, lens=lens)
    _assert(parity.compare(got, ref), str(case))


@pytest.mark.parametrize("variant", [0, 1], ids=["nw", "sw"])
def test_terminal_cell_on_the_border(variant):
    """Single-column / single-row problems and per-pair lengths of 1: for Smith-Waterman the terminal cell then
    lies on the padded border (V = 0 exactly, Vt = 0, E = 0); found by fuzzing when the windowed forward form
    captured an underflowed 0 (-> -inf) for it."""
    for (B, N, M) in [(1, 232, 1), (2, 1, 300), (3, 127, 1), (2, 485, 2)]:
        theta, A = datagen.theta_A(90 + N, B, N, M)
        Z = datagen.normal(91 + N, (B, N, M))
        ref = parity.oracle_all(theta, A, None, Z, variant)
        got = parity.engine_all(theta, A, None, Z, variant)
        assert np.all(np.isfinite(got["Vt"])), (B, N, M)
        _assert(parity.compare(got, ref), f"{(B, N, M)} v{variant}")
    B, N, M = 4, 135, 40
    theta, A = datagen.theta_A(95, B, N, M)
    theta = (5.0 * theta).astype(np.float32)
    Z = datagen.normal(96, (B, N, M))
    lens = np.array([[135, 1], [1, 40], [60, 2], [135, 40]], dtype=np.int32)
    ref = parity.oracle_lens(theta, A, None, Z, variant, lens)
    got = parity.engine_all(theta, A, None, Z, variant, lens=lens)
    assert np.all(np.isfinite(got["Vt"]))
    _assert(parity.compare(got, ref), f"lens v{variant}")


@pytest.mark.parametrize("variant", [0, 1], ids=["nw", "sw"])
def test_forbidden_gaps_minus_infinity(variant):
    """A = -inf (a forbidden gap) is legal in the reference: exp(-inf) = 0 removes the x/y terms (nw.py:10-27).
    The exp-domain forward clamps the exponent instead of forming inf - inf."""
    B, N, M = 2, 90, 130
    theta, A = datagen.theta_A(81, B, N, M)
    A[datagen.uniform(82, (B, N, M)) < 0.05] = -np.inf
    A[0, 10:20, :] = -np.inf
    Z = datagen.normal(83, (B, N, M))
    ref = parity.oracle_all(theta, A, None, Z, variant)
    got = parity.engine_all(theta, A, None, Z, variant)
    assert np.isfinite(got["E"]).all() and np.isfinite(got["Vt"]).all()
    _assert(parity.compare(got, ref))


def test_random_shapes_scales_and_lengths():
    """Seeded fuzz: random (B, N, M), input scales, variant and optional per-pair lengths, all four passes."""
    rng = np.random.default_rng(2024)
    for it in range(60):
        B = int(rng.integers(1, 5))
        N = int(rng.choice([rng.integers(1, 40), rng.integers(40, 200), rng.integers(200, 600)]))
        M = int(rng.choice([rng.integers(1, 40), rng.integers(40, 200), rng.integers(200, 800)]))
        variant = int(rng.integers(0, 2))
        theta, A = datagen.theta_A(10000 + it, B, N, M)
        theta = (theta * float(rng.choice([0.1, 1.0, 5.0]))).astype(np.float32)
        A = (A * float(rng.choice([0.1, 1.0, 10.0])) + float(rng.choice([0.0, 0.0, 0.5]))).astype(np.float32)
        Z = datagen.normal(20000 + it, (B, N, M))
        if rng.integers(0, 2):
            lens = np.stack([rng.integers(1, N + 1, B), rng.integers(1, M + 1, B)], axis=1).astype(np.int32)
            ref = parity.oracle_lens(theta, A, None, Z, variant, lens)
            got = parity.engine_all(theta, A, None, Z, variant, lens=lens)
        else:
            Et = (0.5 + datagen.uniform(30000 + it, (B,))).astype(np.float32)
            ref = parity.oracle_all(theta, A, Et, Z, variant)
            got = parity.engine_all(theta, A, Et, Z, variant)
        _assert(parity.compare(got, ref), f"fuzz {it}: B={B} N={N} M={M} variant={variant}")


def test_max_cols_is_enforced():
    import torch
    from deepblast_amd._engine import get_engine
    eng = get_engine()
    t = torch.zeros(1, 2, eng.max_cols() + 1, device="cuda")
    with pytest.raises(ValueError):
        eng.forward(t, t, 0)


@pytest.mark.parametrize("scale", [(1.0, 1.0), (8.0, 1.0), (30.0, 10.0)], ids=["soft", "theta-x8", "theta-x30-A-x10"])
def test_packed_state_at_the_longest_paths_it_serves(scale):
    """the packed-state limit re-measured for the format in use.  The 20-bit fields serve problems up to
    N + M = 4096 (sdp_api.hip: PACKED_MAX_PATH); their rounding error travels along a path like a random walk, so the longest
    paths are the test: max |dE| against the oracle at 2048 x 2048, and at the headline's 512 x 512 and a long thin 66 x 960
    (the thinnest padded shape that still takes the packed state since round 6: sdp_api.hip exact_for), on soft, steep and
    saturated scores, held to HALF the bound.  (The field width is read off the record stride and printed.)"""
    import torch
    from deepblast_amd._engine import get_engine
    eng = get_engine()
    ts, as_ = scale
    for (B, N, M) in ((2, 2048, 2048), (3, 512, 512), (3, 66, 960)):
        bits = eng.lib.sdp_state_pair_stride(N, M, 0) * 8 // (((N + 63) // 64) * ((M + 126) // 64 * 64) * 64 * 2)
        assert bits == 20
        theta, A = datagen.theta_A(2048 + N, B, N, M)
        theta, A = theta * np.float32(ts), A * np.float32(as_)
        ref = parity.oracle_all(theta, A, None, None, 0, omp=True)
        t, a = torch.from_numpy(theta).cuda(), torch.from_numpy(A).cuda()
        Vt, Q = eng.forward(t, a, 0)
        E = eng.backward(torch.ones(B, device="cuda"), Q, (B, N, M), 0)
        err = parity.abs_err(E.cpu().numpy(), ref["E"])
        print(f"packed {bits}-bit state, {N} x {M}, theta x{ts} A x{as_}: max |dE| = {err:.2e}")
        assert err <= 0.5 * parity.TOL, (N, M, bits, err)
        assert parity.rel_err(Vt.cpu().numpy(), ref["Vt"]) <= parity.TOL


@pytest.mark.parametrize("variant", [0, 1], ids=["nw", "sw"])
def test_more_columns_than_the_sweeps_take_run_transposed(variant):
    """the parity oracle nw.py has no column limit; the engine's sweeps stop at sdp_max_cols() = 2048 (the
    limit of the reference's GPU classes).  The decoders sweep such a problem on the transposed tensors (the recurrence is
    symmetric in its axes): 300 x 4096 against the oracle, first and second order, the reference's gradient quirks intact."""
    import torch
    from deepblast_amd import NeedlemanWunschDecoder, SmithWatermanDecoder
    B, N, M = 2, 300, 4096
    theta, A = datagen.theta_A(4096 + variant, B, N, M)
    Z = datagen.normal(4100, (B, N, M))
    dec = (NeedlemanWunschDecoder, SmithWatermanDecoder)[variant]("softmax")
    t = torch.from_numpy(theta).cuda().requires_grad_()
    a = torch.from_numpy(A).cuda().requires_grad_()
    with torch.no_grad():
        vt = dec(t, a)
    aln = dec.decode(t, a)
    assert aln.shape == (B, N, M)
    (aln * torch.from_numpy(Z).cuda()).sum().backward()
    torch.cuda.synchronize()
    ref = parity.oracle_all(theta, A, None, Z, variant, omp=True)
    # (plain max-abs for NW: 3.7e-5 with max|Ed_ref| = 8.1; SW: 1.05e-4 with max|Ed_ref| = 5.6 on this 4395-step problem --
    #  the scaled figure, 1.9e-5, is what is held there)
    errs = parity.compare({"Vt": vt.cpu().numpy(), "E": aln.detach().cpu().numpy(), "Ed": t.grad.cpu().numpy(), "Vtd": ref["Vtd"]}, ref,
                          plain=variant == 0)
    print(f"\ntransposed sweep {N}x{M} variant={variant}: {errs}  max|Ed_ref| = {np.abs(ref['Ed']).max():.2f}")
    _assert(errs, f"transposed sweep {N}x{M} variant={variant}")
    assert parity.abs_err(t.grad.cpu().numpy(), ref["Ed"]) <= PLAIN_CEILING   # (the plain figure under its explicit ceiling, both variants)
    assert a.grad is None      # second order: no gradient for A (nw.py:386)
    # first order through forward(): E in theta.grad, the pass-through "gradient" A in A.grad (nw.py:337-339,355)
    t2 = torch.from_numpy(theta).cuda().requires_grad_()
    a2 = torch.from_numpy(A).cuda().requires_grad_()
    dec(t2, a2).sum().backward()
    assert parity.abs_err(t2.grad.cpu().numpy(), ref["E"]) <= parity.TOL and torch.equal(a2.grad, a2.detach())
    # per-pair lengths ride along (their columns swapped)
    lens = np.array([[300, 4096], [123, 3000]], np.int32)
    refl = parity.oracle_lens(theta, A, None, None, variant, lens)
    t3 = torch.from_numpy(theta).cuda().requires_grad_()
    v3 = dec(t3, torch.from_numpy(A).cuda(), torch.from_numpy(lens).cuda())
    v3.sum().backward()
    assert parity.rel_err(v3.detach().cpu().numpy(), refl["Vt"]) <= parity.TOL and parity.abs_err(t3.grad.cpu().numpy(), refl["E"]) <= parity.TOL
    # both sides beyond the limit: still refused
    big = torch.zeros(1, 2049, 2049, device="cuda")
    with pytest.raises(ValueError):
        dec(big, big)


def test_fuzz_many_pairs_small_odd_shapes():
    """The corner the line-aligned staging of round 2 first got wrong (found by tools/fuzz2.py, not by the suite): full
    batches (>= 128 pairs: throughput builds) of small matrices whose planes start at arbitrary float offsets (N*M odd),
    one partial strip, M below a chunk.  60 seeded cases, all four sweeps, NW and SW, with and without lengths."""
    rng = np.random.default_rng(4242)
    for it in range(60):
        B, N, M = int(rng.integers(128, 300)), int(rng.integers(1, 90)), int(rng.integers(1, 90))
        variant = int(rng.integers(0, 2))
        theta, A = datagen.theta_A(70000 + it, B, N, M)
        theta = (theta * float(rng.choice([0.01, 1.0, 8.0]))).astype(np.float32)
        A = (A * float(rng.choice([0.0, 1.0, 10.0])) + float(rng.choice([0.0, 0.5, -3.0]))).astype(np.float32)
        Z = datagen.normal(71000 + it, (B, N, M))
        if rng.integers(0, 3) == 0:
            lens = np.stack([rng.integers(1, N + 1, B), rng.integers(1, M + 1, B)], axis=1).astype(np.int32)
            ref = parity.oracle_lens(theta, A, None, Z, variant, lens)
            got = parity.engine_all(theta, A, None, Z, variant, lens=lens)
        else:
            ref = parity.oracle_all(theta, A, None, Z, variant)
            got = parity.engine_all(theta, A, None, Z, variant)
        _assert(parity.compare(got, ref), f"case {it}: {(B, N, M, variant)}")


def _fuzz2_case(rng, it):
    """One case of tools/fuzz2.py's mixed family: steep and flat scores, forbidden gaps, long rows, tiny shapes, many
    pairs, per-pair lengths, ZA and Et."""
    kind = int(rng.integers(0, 5))
    if kind == 0: B, N, M = int(rng.integers(1, 4)), int(rng.integers(1, 8)), int(rng.integers(1, 2049))
    elif kind == 1: B, N, M = int(rng.integers(1, 4)), int(rng.integers(1, 1500)), int(rng.integers(1, 8))
    elif kind == 2: B, N, M = int(rng.integers(1, 300)), int(rng.integers(1, 90)), int(rng.integers(1, 90))
    else: B, N, M = int(rng.integers(1, 4)), int(rng.integers(1, 700)), int(rng.integers(1, 900))
    variant = int(rng.integers(0, 2))
    theta, A = datagen.theta_A(50000 + it, B, N, M)
    ts = float(rng.choice([0.01, 1.0, 8.0, 30.0])); as_ = float(rng.choice([0.0, 1.0, 10.0, 40.0])); ao = float(rng.choice([0.0, 0.5, -3.0]))
    theta = (theta * ts - float(rng.choice([0.0, 0.0, 2.0]))).astype(np.float32)
    A = (A * as_ + ao).astype(np.float32)
    if rng.integers(0, 6) == 0:
        A[rng.random(A.shape) < 0.2] = -np.inf
    Z = datagen.normal(60000 + it, (B, N, M))
    lens = ZA = Et = None
    if rng.integers(0, 2):
        lens = np.stack([rng.integers(1, N + 1, B), rng.integers(1, M + 1, B)], axis=1).astype(np.int32)
    else:
        ZA = datagen.normal(70000 + it, (B, N, M)) if rng.integers(0, 3) == 0 else None
        Et = rng.normal(size=B).astype(np.float32) if rng.integers(0, 3) == 0 else None
    return dict(theta=theta, A=A, Z=Z, variant=variant, lens=lens, ZA=ZA, Et=Et, tag=(B, N, M, variant, ts, as_, ao))


@pytest.mark.parametrize("part", range(4))
def test_fuzz2_hundred_seeded_cases(part):
    """100 seeded cases of the harsher fuzz family (tools/fuzz2.py; it found the one kernel bug of round 2 that the
    suite had missed), 25 per test: all four sweeps at the ordinary bound.  A second-order result over the bound is
    accepted only where the engine agrees with the float64 reference to 2e-5 and the reference's own fp32 and float64
    runs differ by that much (INTEGRATION.md states where that can happen)."""
    rng = np.random.default_rng(31337 + part)
    for it in range(part * 25, part * 25 + 25):
        c = _fuzz2_case(rng, it)
        if c["lens"] is not None:
            ref = parity.oracle_lens(c["theta"], c["A"], None, c["Z"], c["variant"], c["lens"])
            got = parity.engine_all(c["theta"], c["A"], None, c["Z"], c["variant"], lens=c["lens"])
        else:
            ref = parity.oracle_all(c["theta"], c["A"], c["Et"], c["Z"], c["variant"], ZA=c["ZA"])
            got = parity.engine_all(c["theta"], c["A"], c["Et"], c["Z"], c["variant"], ZA=c["ZA"])
        e = parity.compare(got, ref)
        first = {k: e[k] for k in ("Vt", "E", "Ex", "Vtx")}
        _assert(first, f"fuzz2 {it} {c['tag']}")
        second = max(e["Ed"], e["Vtd"])
        # the reference-rounding mode takes every case at a fiftieth of the bound, no exemption: every fifth case here (the
        # mode is unoptimised), and every case whose fast-path second-order result needs the cross-check below
        if it % 5 == 0 or not (np.isfinite(second) and second <= parity.TOL):
            if not np.isinf(c["A"]).any():   # (A = -inf: inf - inf in the reference's own arithmetic, NaN on both sides)
                eref = parity.compare(parity.engine_ref(c["theta"], c["A"], c["Et"], c["Z"], c["variant"], lens=c["lens"], ZA=c["ZA"]), ref)
                assert max(eref.values()) <= 0.02 * parity.TOL, (it, c["tag"], eref)
        if not (np.isfinite(second) and second <= parity.TOL):
            assert c["lens"] is None, (it, c["tag"], e)
            f8 = lambda x: None if x is None else x.astype(np.float64)
            r64 = parity.oracle_all(f8(c["theta"]), f8(c["A"]), f8(c["Et"]), f8(c["Z"]), c["variant"], ZA=f8(c["ZA"]))
            e64, noise = parity.compare(got, r64), parity.compare(ref, r64)
            assert max(e64["Ed"], e64["Vtd"]) <= 0.2 * parity.TOL and max(noise["Ed"], noise["Vtd"]) >= 0.9 * second, (it, c["tag"], e, e64, noise)


@pytest.mark.parametrize("case", [(5, 1024, 1024, 0, False), (3, 1000, 700, 1, False), (2, 1990, 1200, 0, False), (40, 700, 333, 1, True),
                                  (256, 640, 200, 0, True), (7, 321, 1500, 0, False), (300, 900, 130, 0, True)],
                         ids=lambda c: "x".join(str(int(v)) for v in c))
def test_pairs_spread_over_several_workgroups(case):
    """Long pairs cut into parts of four strips, each part a workgroup on its own CU, the boundary between two parts
    crossing CUs through 8-byte granules in global memory (sdp_kernels.hip: PARTS, bridge).  The library does that where
    it pays (sdp_api.hip::plan); here the experiments build forces it in both sweeps for every case, and the results must
    be bit-identical to the one-workgroup-per-pair schedule -- forward and backward, packed and exact state, per-pair
    lengths with the dispatch map -- and in parity with the oracle through the shipped library."""
    import ctypes
    import torch
    from deepblast_amd import _lib, build
    from deepblast_amd._engine import get_engine
    B, N, M, variant, use_lens = case
    eng = get_engine()
    theta, A = datagen.theta_A(91000 + N, B, N, M)
    theta = (theta * (3.0 if B < 10 else 1.0)).astype(np.float32)
    lens = None
    if use_lens:
        lens = datagen.lengths(91001, B, 1, N)
        lens[:, 1] = np.minimum(lens[:, 1] * M // N + 1, M)
        lens[0] = (N, M)
        lens[1] = (N, 1)
        lens[2] = (257, M)
    ref = parity.oracle_lens(theta, A, None, None, variant, lens, threads=16) if use_lens else parity.oracle_all(theta, A, None, None, variant, omp=True)
    t, a = torch.from_numpy(theta).cuda(), torch.from_numpy(A).cuda()
    et = torch.from_numpy((0.5 + datagen.uniform(91002, (B,))).astype(np.float32)).cuda()
    ln = None if lens is None else torch.from_numpy(lens).cuda()
    ones = torch.ones(B, device="cuda")
    out = {}
    for exact in (False, True):
        Vt, Q = eng.forward(t, a, variant, ln, exact_state=exact)
        E = eng.backward(ones, Q, (B, N, M), variant, ln, exact_state=exact)
        E2 = eng.backward(et, Q, (B, N, M), variant, ln, exact_state=exact)
        torch.cuda.synchronize()
        assert eng.check_device()[0] == 0
        errs = parity.compare({"Vt": Vt.cpu().numpy(), "E": E.cpu().numpy()}, ref)
        _assert(errs, f"parts {case} exact={exact}")
        out[exact] = (Vt, E, E2)
    # the experiments build: parts forced on against parts switched off, both with the 4-wave throughput kernels (what the
    # parts run on; the 8-wave latency builds a small batch would otherwise take cut the columns into shorter chunks and
    # may round a block differently): same bits
    exp = _lib.load_path(build.EXP_OUT)
    stream = torch.cuda.current_stream().cuda_stream
    info0 = (ctypes.c_int32 * 4)()
    exp.sdp_device_status(torch.cuda.current_device(), info0)
    timeouts_before = info0[0]
    for exact in (False, True):
        flag = 0x100 if exact else 0
        nbytes = (exp.sdp_state_d_bytes if exact else exp.sdp_state_bytes)(B, N, M)
        res = []
        for mask, waves in ((64, 4), (512, 0)):
            exp.sdp_set_debug(mask)
            st = torch.empty(nbytes // 4, device="cuda")
            vt = torch.empty(B, device="cuda")
            E = torch.empty(B, N, M, device="cuda")
            lp = None if ln is None else ln.data_ptr()
            assert exp.sdp_forward_f32(t.data_ptr(), a.data_ptr(), st.data_ptr(), vt.data_ptr(), B, N, M, lp, variant | flag | (waves << 12), 0, stream) == 0
            assert exp.sdp_backward_f32(et.data_ptr(), st.data_ptr(), E.data_ptr(), B, N, M, lp, variant | flag | (waves << 12), 0, stream) == 0
            exp.sdp_set_debug(0)
            torch.cuda.synchronize()
            info = (ctypes.c_int32 * 4)()   # (info[0] counts since the library was loaded: other tests force time-outs in this build)
            exp.sdp_device_status(torch.cuda.current_device(), info)
            assert info[0] == timeouts_before, (case, exact, mask, list(info))
            res.append((vt, E))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), (case, exact)
        if all(eng.lib.sdp_plan_parts(p_, B, N, M, int(use_lens), int(exact), 256) == 4 for p_ in (0, 1)):   # what the library does by itself
            assert torch.equal(res[1][0], out[exact][0]) and torch.equal(res[1][1], out[exact][2]), (case, exact)


@pytest.mark.parametrize("offset", [0, 1, 2, 3, 7, 19], ids=lambda o: f"plane+{o}floats")
def test_soak_case_of_round_5_neighbour_frames_far_apart(offset):
    """Found by the 1200-case soak of round 5 (tools/fuzz2.py case 1172, pair 37 of 133: Smith-Waterman, theta x 8, A = 0, 71 x 81):
    the windowed forward form rescaled a neighbour's value as ua * 2^(Rn - R), and the factor alone underflowed to zero where lane
    0's frame -- the producer strip's -- lay far below lane 1's: the `up` weight of cell (65, 79) came out 0 instead of 0.975, Vt
    was off by 0.033 and E by 5e-3.  Which blocks run in the windowed form depends on what lies beside the matrix in memory, so the
    case needs its plane to start 3 floats off a 16-byte boundary: every offset is run.  (tests/golden/r5_soak_case1172_pair37.npz
    holds the pair's scores as the fuzz generated them; expected values: the oracle.)"""
    import torch
    from deepblast_amd._engine import get_engine
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "r5_soak_case1172_pair37.npz"))
    theta, A, Et, variant = d["theta"], d["A"], d["Et"], int(d["variant"])
    ref = parity.oracle_all(theta, A, Et, None, variant, omp=False)
    lib = get_engine().lib
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(0).cuda_stream
    B, N, M = theta.shape
    for waves in (0, 1, 4):
        buf_t = torch.zeros(theta.size + 64, device=dev)
        buf_a = torch.zeros(theta.size + 64, device=dev)
        t = buf_t[offset:offset + theta.size].view(B, N, M)
        a = buf_a[offset:offset + theta.size].view(B, N, M)
        t.copy_(torch.from_numpy(theta)), a.copy_(torch.from_numpy(A))
        et = torch.from_numpy(Et).to(dev)
        st = torch.empty(lib.sdp_state_bytes(B, N, M) // 4 + 64, device=dev)
        vt = torch.empty(B, device=dev)
        E = torch.full((B, N, M), float("nan"), device=dev)
        assert lib.sdp_forward_f32(t.data_ptr(), a.data_ptr(), st.data_ptr(), vt.data_ptr(), B, N, M, None, variant | (waves << 12), 0, stream) == 0
        assert lib.sdp_backward_f32(et.data_ptr(), st.data_ptr(), E.data_ptr(), B, N, M, None, variant | (waves << 12), 0, stream) == 0
        torch.cuda.synchronize()
        assert parity.rel_err(vt.cpu().numpy(), ref["Vt"]) <= 1e-6, (offset, waves, float(vt[0]), float(ref["Vt"][0]))
        assert parity.abs_err(E.cpu().numpy(), ref["E"]) <= parity.TOL, (offset, waves)


@pytest.mark.parametrize("shape", [(2, 2, 1772), (1, 1772, 3), (2, 16, 2048), (1, 31, 600)], ids=lambda s: "x".join(map(str, s)))
def test_thin_long_problems_meet_the_bound_with_room(shape):
    """Round 5: on thin long problems the packed state's rounding does not average out (flat scores, positive gap scores: 1.17e-4
    at 2 x 1772 in the round's soak, 1.0e-4 at 2 x 2048 in tools/thin_probe.py); they take the exact state now: a fifth of the bound."""
    B, N, M = shape
    for variant in (0, 1):
        theta, A = datagen.theta_A(8800 + N, B, N, M)
        theta = (theta * 0.01).astype(np.float32)
        A = (A + 0.5).astype(np.float32)
        ref = parity.oracle_all(theta, A, None, None, variant, omp=False)
        got = parity.engine_all(theta, A, None, None, variant)
        errs = parity.compare(got, ref)
        assert errs["E"] <= 0.2 * parity.TOL and errs["Vt"] <= 0.2 * parity.TOL, (shape, variant, errs)


@pytest.mark.parametrize("which", ["a", "b"])
def test_soak_cases_of_round_5_cells_that_nothing_reaches(which):
    """Found by the last soak of round 5 (tools/fuzz2.py 2000 11 / 12: 3 of 4000 cases): thin problems (3 rows) with a fifth of
    their gap scores -inf have cells that nothing reaches -- every way in is a forbidden gap and the diagonal predecessor is
    unreachable itself.  Their sum of three was 0, their weights 0 * (1 / 0) = NaN in the exact state; the first-order sweeps
    never noticed (E is 0 there), the adjoint sweeps spread the NaN over the whole pair (Vtd = NaN, Ed = NaN).  The reference
    works with finite -1e10 borders and gets weights (0, 1, 0) there.  Fixture: the two pairs as the fuzz generated them
    (one with a per-pair Et, one with a ZA seed); expected: the oracle, every output finite and inside the bound."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "r5_soak_unreachable_cells.npz"))
    theta, A, Z, variant = d[f"theta_{which}"], d[f"A_{which}"], d[f"Z_{which}"], int(d[f"variant_{which}"])
    Et = d["Et_a"] if which == "a" else None
    ZA = d["ZA_b"] if which == "b" else None
    assert np.isneginf(A).sum() > 100
    ref = parity.oracle_all(theta, A, Et, Z, variant, ZA=ZA, omp=False)
    got = parity.engine_all(theta, A, Et, Z, variant, ZA=ZA)
    for k in ("Vt", "E", "Ed", "Vtd"):
        assert np.isfinite(got[k]).all(), k
    _assert(parity.compare(got, ref), f"unreachable cells, case {which}")
