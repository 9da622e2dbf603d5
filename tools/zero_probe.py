"""Exact-zero skip of the fp32 backward sweep (sdp_kernels.hip, "exact zeros"): experiments build, same box, same state,
debug bit 4096 runs the steps of all-zero chunks too.  usage: zero_probe.py [B N M]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import datagen, gpu_tune
B, N, M = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (256, 512, 512)
exp = gpu_tune.load(os.path.join(ROOT, "deepblast_amd", "libsdp_hip_exp.so"))
stream = torch.cuda.current_stream().cuda_stream
th, A = datagen.theta_A(1, B, N, M)
for name, scale, exact in (("bench data (theta ~ U[0,1), A ~ -U[0,1))", 1.0, 0), ("theta x 4", 4.0, 0), ("theta x 16 (peaked)", 16.0, 0), ("theta x 0.25 (flat)", 0.25, 0),
                           ("bench data, float2 state", 1.0, 0x100)):
    t, a = torch.from_numpy(th * np.float32(scale)).cuda(), torch.from_numpy(A).cuda()
    st = torch.empty(exp.sdp_state_bytes_v(B, N, M, exact) // 4 + 1, device="cuda")
    vt = torch.empty(B, device="cuda"); et = torch.ones(B, device="cuda"); E = torch.empty(B, N, M, device="cuda")
    assert exp.sdp_forward_f32(t.data_ptr(), a.data_ptr(), st.data_ptr(), vt.data_ptr(), B, N, M, None, exact, 0, stream) == 0
    g = lambda: exp.sdp_backward_f32(et.data_ptr(), st.data_ptr(), E.data_ptr(), B, N, M, None, exact, 0, stream)
    res = {}
    for rep in range(3):
        for mask in (0, 4096):
            gpu_tune.set_debug(exp, mask)
            assert g() == 0
            res.setdefault(mask, []).append(gpu_tune.timeit(g))
    gpu_tune.set_debug(exp, 0)
    zf = float((E == 0).float().mean())
    print(f"{name:45s} zero cells {zf:.3f}  bwd skip {np.median(res[0]):.1f} us  no skip {np.median(res[4096]):.1f} us", flush=True)
