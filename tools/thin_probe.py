"""Packed (20-bit) state on THIN problems with flat scores and positive gap scores: max |dE| against the oracle, packed / exact
state, by shape (the soak of round 5 met 1.17e-4 at 2 x 1772, Smith-Waterman, theta x 0.01, A ~ N(0.5, 1))."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen, parity
from deepblast_amd._engine import get_engine
lib = get_engine().lib
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(0).cuda_stream
def run(theta, A, variant, flag):
    B, N, M = theta.shape
    t, a = torch.from_numpy(theta).to(dev), torch.from_numpy(A).to(dev)
    et = torch.ones(B, device=dev)
    st = torch.empty(max(lib.sdp_state_bytes(B, N, M), lib.sdp_state_d_bytes(B, N, M)) // 4 + 64, device=dev); vt = torch.empty(B, device=dev); E = torch.zeros(B, N, M, device=dev)
    assert lib.sdp_forward_f32(t.data_ptr(), a.data_ptr(), st.data_ptr(), vt.data_ptr(), B, N, M, None, variant | flag, 0, stream) == 0
    assert lib.sdp_backward_f32(et.data_ptr(), st.data_ptr(), E.data_ptr(), B, N, M, None, variant | flag, 0, stream) == 0
    torch.cuda.synchronize()
    return E.cpu().numpy()
for (ts, a_s, a_o) in ((0.01, 1.0, 0.5), (1.0, 1.0, 0.5), (0.01, 1.0, 0.0), (1.0, 1.0, 0.0), (0.01, 0.0, 0.5)):
    print(f"theta x {ts}, A x {a_s} + {a_o}: worst max|dE| over 6 seeds x NW/SW, packed (exact)")
    for N in (1, 2, 3, 4, 8, 16, 32, 64):
        row = []
        for M in (256, 512, 1024, 1772, 2048):
            wp = wx = 0.0
            for seed in range(6):
                for variant in (0, 1):
                    theta, A = datagen.theta_A(123000 + 17 * seed + N, 2, N, M)
                    theta = (theta * ts).astype(np.float32); A = (A * a_s + a_o).astype(np.float32)
                    ref = parity.oracle_all(theta, A, None, None, variant, omp=False)
                    wp = max(wp, float(np.abs(run(theta, A, variant, 0) - ref["E"]).max()))
                    wx = max(wx, float(np.abs(run(theta, A, variant, 0x100) - ref["E"]).max()))
            row.append(f"{M}: {wp:.1e} ({wx:.1e})")
        print(f"   N={N:3d}  " + "   ".join(row), flush=True)
