"""One dumped fuzz2 case (FUZZ_ONLY=<it> FUZZ_DUMP=<npz> python tools/fuzz2.py <n>) under the microscope: per-pair errors of Vt / E
against the oracle, where the worst cell lies, and the same case with the zero-chunk skip off, forced wave counts, the exact state.
usage: fuzz2_case.py <npz>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import parity
from oracle import oracle
from deepblast_amd import _lib
from deepblast_amd._engine import get_engine

d = np.load(sys.argv[1])
theta, A, variant = d["theta"], d["A"], int(d["variant"])
Et = d["Et"] if d["Et"].size else None
lens = d["lens"] if d["lens"].size else None
if os.environ.get("PAIR"):   # PAIR=<b>: that pair alone (B = 1)
    b0 = int(os.environ["PAIR"])
    theta, A = theta[b0:b0 + 1].copy(), A[b0:b0 + 1].copy()
    Et = None if Et is None else Et[b0:b0 + 1].copy()
    lens = None if lens is None else lens[b0:b0 + 1].copy()
B, N, M = theta.shape
ref = parity.oracle_all(theta, A, Et, None, variant) if lens is None else parity.oracle_lens(theta, A, None, None, variant, lens)
lib = get_engine().lib
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(0).cuda_stream
t, a = torch.from_numpy(theta).to(dev), torch.from_numpy(A).to(dev)
et = torch.from_numpy(Et if Et is not None else np.ones(B, np.float32)).to(dev)
lp = None
if lens is not None:
    lt = torch.from_numpy(lens).to(dev); lp = lt.data_ptr()
print(f"case {sys.argv[1]}: B={B} N={N} M={M} variant={variant} theta in [{theta.min():.3g}, {theta.max():.3g}] A in [{A.min():.3g}, {A.max():.3g}] Et {'given' if Et is not None else 'ones'}")
for name, ff, fb in [("default", 0, 0), ("no zero skip", 0, _lib.SDP_NO_ZERO_SKIP), ("exact state", 0x100, 0x100), ("1 wave", 1 << 12, 1 << 12), ("2 waves", 2 << 12, 2 << 12),
                     ("4 waves", 4 << 12, 4 << 12), ("8 waves", 8 << 12, 8 << 12), ("fwd auto, bwd 4 waves", 0, 4 << 12), ("fwd 4 waves, bwd auto", 4 << 12, 0)]:
    st = torch.empty(max(lib.sdp_state_bytes(B, N, M), lib.sdp_state_d_bytes(B, N, M)) // 4 + 64, device=dev)
    vt = torch.empty(B, device=dev)
    E = torch.full((B, N, M), float("nan"), device=dev)
    rc1 = lib.sdp_forward_f32(t.data_ptr(), a.data_ptr(), st.data_ptr(), vt.data_ptr(), B, N, M, lp, variant | ff, 0, stream)
    rc2 = lib.sdp_backward_f32(et.data_ptr(), st.data_ptr(), E.data_ptr(), B, N, M, lp, variant | fb, 0, stream)
    torch.cuda.synchronize()
    if rc1 or rc2:
        print(f"  {name:24s} rc {rc1} {rc2}: {lib.sdp_last_error_string()}"); continue
    Eg, Vg = E.cpu().numpy(), vt.cpu().numpy()
    dE = np.abs(Eg - ref["E"]).reshape(B, -1).max(axis=1)
    dV = np.abs(Vg - ref["Vt"]) / np.maximum(1.0, np.abs(ref["Vt"]))
    bad = np.nonzero(dE > 1e-4)[0]
    b = int(dE.argmax()); i, j = np.unravel_index(np.abs(Eg[b] - ref["E"][b]).argmax(), (N, M))
    print(f"  {name:24s} max|dE| {dE.max():.3e} (pair {b}, cell {i},{j}: got {Eg[b, i, j]:.6g} want {ref['E'][b, i, j]:.6g}, Et {et[b].item():.4g})  pairs over 1e-4: {len(bad)} {bad[:12].tolist()}  max rel dVt {dV.max():.2e}  nan {int(np.isnan(Eg).sum())}")
    if name == "default" and len(bad):
        b = int(bad[0])
        err = np.abs(Eg[b] - ref["E"][b])
        rows = np.nonzero(err.max(axis=1) > 1e-4)[0]; cols = np.nonzero(err.max(axis=0) > 1e-4)[0]
        print(f"     pair {b}: cells over 1e-4: {int((err > 1e-4).sum())}, rows {rows.min()}..{rows.max()}, cols {cols.min()}..{cols.max()}; sum E got {Eg[b].sum():.5f} want {ref['E'][b].sum():.5f}; Vt got {Vg[b]:.6f} want {ref['Vt'][b]:.6f}")
