"""Fuzz aimed at the forward sweep's windowed form: steep scores at the edge of its range (|theta| log2(e) <= 12),
flat gap scores, two to four strips, few columns beyond a block boundary, Smith-Waterman and Needleman-Wunsch, and EVERY alignment
of the planes (the range test of a block also sees what lies beside the matrix in memory, so which blocks run windowed depends on
it).  First order against the oracle; forced wave counts so that the K = 32 builds run whatever the batch.
usage: wf_fuzz.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import datagen, parity
from deepblast_amd._engine import get_engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2025)
lib = get_engine().lib
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(0).cuda_stream
worst, bad = 0.0, 0
for it in range(n):
    B = int(rng.integers(1, 4)); N = int(rng.integers(60, 270)); M = int(rng.integers(17, 330))
    variant = int(rng.integers(0, 2))
    ts = float(rng.choice([4.0, 8.0, 8.3, 11.0])); a_s = float(rng.choice([0.0, 0.0, 0.05, 1.0])); a_o = float(rng.choice([0.0, 0.0, -0.5, 0.3]))
    theta, A = datagen.theta_A(90000 + it, B, N, M)
    theta = (theta * ts / max(1e-6, float(np.abs(theta).max())) * float(rng.choice([1.0, 1.0, 0.5]))).astype(np.float32) if rng.integers(0, 2) else (theta * ts).astype(np.float32)
    A = (A * a_s + a_o).astype(np.float32)
    Et = (0.5 + datagen.uniform(91000 + it, (B,))).astype(np.float32)
    ref = parity.oracle_all(theta, A, Et, None, variant, omp=False)
    for off in range(4):
        for waves in (1, 4):
            bt = torch.zeros(theta.size + 64, device=dev); ba = torch.zeros(theta.size + 64, device=dev)
            t = bt[off:off + theta.size].view(B, N, M); a = ba[off:off + theta.size].view(B, N, M)
            t.copy_(torch.from_numpy(theta)); a.copy_(torch.from_numpy(A))
            et = torch.from_numpy(Et).to(dev)
            st = torch.empty(lib.sdp_state_bytes(B, N, M) // 4 + 64, device=dev); vt = torch.empty(B, device=dev); E = torch.full((B, N, M), float("nan"), device=dev)
            assert lib.sdp_forward_f32(t.data_ptr(), a.data_ptr(), st.data_ptr(), vt.data_ptr(), B, N, M, None, variant | (waves << 12), 0, stream) == 0
            assert lib.sdp_backward_f32(et.data_ptr(), st.data_ptr(), E.data_ptr(), B, N, M, None, variant | (waves << 12), 0, stream) == 0
            torch.cuda.synchronize()
            e = max(parity.abs_err(E.cpu().numpy(), ref["E"]), parity.rel_err(vt.cpu().numpy(), ref["Vt"]))
            e = e if np.isfinite(e) else 9e9
            worst = max(worst, e)
            if e > parity.TOL:
                bad += 1
                print(f"it={it} {(B, N, M, variant)} theta scale {ts} A*{a_s}+{a_o} offset {off} waves {waves}: {e:.3e}", flush=True)
print(f"{n} cases x 4 plane offsets x 2 wave counts: worst first-order error {worst:.3e}, {bad} over 1e-4")
