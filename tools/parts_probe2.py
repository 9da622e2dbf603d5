"""configs[2] (per-pair lengths) through the shipped library and the experiments build: forward / backward launch times back to back and alternating."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import datagen, gpu_tune
B = 256
ln3 = datagen.lengths(2, B, 64, 1024)
N, M = int(ln3[:, 0].max()), int(ln3[:, 1].max())
if "aligned" in sys.argv:
    N = M = 1024   # rows of the padded batch on 128-byte lines: the kernels without per-row offsets
if "round32" in sys.argv:   # column tails of E then start on 128-byte lines (with `aligned`)
    ln3[:, 1] = np.minimum((ln3[:, 1] + 31) // 32 * 32, M)
theta, A = datagen.theta_A(2, B, N, M)
t, a = torch.from_numpy(theta).cuda(), torch.from_numpy(A).cuda()
et = torch.ones(B, device="cuda")
lens = torch.from_numpy(ln3).cuda()
stream = torch.cuda.current_stream().cuda_stream
import glob
for name in [os.path.join(ROOT, "deepblast_amd", n) for n in ("libsdp_hip.so", "libsdp_hip_exp.so")] + sorted(glob.glob(os.path.join(ROOT, "build_variants", "*.so"))):
    lib = gpu_tune.load(name)
    name = os.path.basename(name)
    st = torch.empty(lib.sdp_state_bytes(B, N, M) // 4, device="cuda")
    vt = torch.empty(B, device="cuda"); E = torch.empty(B, N, M, device="cuda")
    f = lambda: lib.sdp_forward_f32(t.data_ptr(), a.data_ptr(), st.data_ptr(), vt.data_ptr(), B, N, M, lens.data_ptr(), 0, 0, stream)
    g = lambda: lib.sdp_backward_f32(et.data_ptr(), st.data_ptr(), E.data_ptr(), B, N, M, lens.data_ptr(), 0, 0, stream)
    assert f() == 0 and g() == 0
    torch.cuda.synchronize()
    print(name, f"fwd {gpu_tune.timeit(f):.1f} us  bwd {gpu_tune.timeit(g):.1f} us  fwd;bwd {gpu_tune.timeit(lambda: (f(), g())):.1f} us", flush=True)
