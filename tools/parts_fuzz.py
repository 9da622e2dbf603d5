"""Fuzz of the parts schedule (a pair spread over several workgroups): random shapes, per-pair lengths (degenerate ones
included), NW / SW, packed / exact state -- parts forced on (experiments build, sdp_set_debug(512)) against one workgroup
per pair with the same 4-wave throughput kernels: Vt and E must agree bit for bit, and no hand-off may time out.
usage: python tools/parts_fuzz.py [cases] [seed]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import gpu_tune
ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
exp = gpu_tune.load(os.path.join(ROOT, "deepblast_amd", "libsdp_hip_exp.so"))
stream = torch.cuda.current_stream().cuda_stream
info = (ctypes.c_int32 * 4)()
exp.sdp_device_status(0, info)
t0 = info[0]
bad = 0
for it in range(ncases):
    N = int(rng.integers(257, 2100))
    M = int(rng.choice([rng.integers(1, 64), rng.integers(64, 700), rng.integers(700, 2048)]))
    B = int(rng.integers(1, max(2, min(300, (1 << 27) // (N * M)))))
    variant = int(rng.integers(0, 2))
    exact = bool(rng.integers(0, 2))
    use_lens = bool(rng.integers(0, 3))        # two thirds with per-pair lengths
    scale = float(rng.choice([0.3, 1.0, 4.0, 20.0]))
    theta = torch.from_numpy((rng.standard_normal((B, N, M)) * scale).astype(np.float32)).cuda()
    A = torch.from_numpy((-np.abs(rng.standard_normal((B, N, M))) * float(rng.choice([0.1, 1.0, 5.0]))).astype(np.float32)).cuda()
    et = torch.from_numpy((0.5 + rng.random(B)).astype(np.float32)).cuda()
    lp = None
    if use_lens:
        ln = np.stack([rng.integers(1, N + 1, B), rng.integers(1, M + 1, B)], axis=1).astype(np.int32)
        ln[rng.integers(0, B)] = (N, M)
        if B > 2:
            ln[rng.integers(0, B)] = (N, 1)
            ln[rng.integers(0, B)] = (1, M)
        lens = torch.from_numpy(ln).cuda()
        lp = lens.data_ptr()
    flag = 0x100 if exact else 0
    nbytes = (exp.sdp_state_d_bytes if exact else exp.sdp_state_bytes)(B, N, M)
    res = []
    for mask, waves in ((64, 4), (512, 0)):
        gpu_tune.set_debug(exp, mask)
        st = torch.empty(nbytes // 4, device="cuda")
        vt = torch.empty(B, device="cuda")
        E = torch.empty(B, N, M, device="cuda")
        r1 = exp.sdp_forward_f32(theta.data_ptr(), A.data_ptr(), st.data_ptr(), vt.data_ptr(), B, N, M, lp, variant | flag | (waves << 12), 0, stream)
        r2 = exp.sdp_backward_f32(et.data_ptr(), st.data_ptr(), E.data_ptr(), B, N, M, lp, variant | flag | (waves << 12), 0, stream)
        torch.cuda.synchronize()
        res.append((r1, r2, vt, E))
    gpu_tune.set_debug(exp, 0)
    parts = [exp.sdp_plan_parts(p, B, N, M, int(use_lens), int(exact), 256) for p in (0, 1)]
    ok = res[0][0] == 0 and res[0][1] == 0 and res[1][0] == 0 and res[1][1] == 0 and torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][3], res[1][3])
    exp.sdp_device_status(0, info)
    if not ok or info[0] != t0:
        bad += 1
        print(f"it={it} B={B} N={N} M={M} variant={variant} exact={exact} lens={use_lens} scale={scale}: MISMATCH rc={[r[:2] for r in res]} timeouts={info[0] - t0}"
              f" dVt={float((res[0][2] - res[1][2]).abs().max()):.2e} dE={float((res[0][3] - res[1][3]).abs().max()):.2e}", flush=True)
        t0 = info[0]
    del theta, A, res
print(f"{ncases} cases: {bad} mismatches (parts forced on vs one workgroup per pair, bit for bit)")
