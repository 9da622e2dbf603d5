"""BASELINE configs[2] (256 pairs, n, m ~ U[64, 1024], padded) through every library build (main + build_variants/*): forward and
backward sweep with per-pair lengths, interleaved, us per launch (median of REPS bursts of 10).  usage: lens_ab.py [nofill]"""
import glob, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpu_tune
sys.path.insert(0, os.path.join(gpu_tune.ROOT, "tests"))
import datagen
libs = {"main": os.path.join(gpu_tune.ROOT, "deepblast_amd", "libsdp_hip.so")}
if "noroute" in sys.argv:
    libs["exp"] = os.path.join(gpu_tune.ROOT, "deepblast_amd", "libsdp_hip_exp.so")   # the experiments build: has sdp_set_debug
for p in sorted(glob.glob(os.path.join(gpu_tune.ROOT, "build_variants", "libsdp_*.so"))):
    libs[os.path.basename(p)[7:-3]] = p
L = {k: gpu_tune.load(v) for k, v in libs.items()}
B = 256
ln = datagen.lengths(2, B, 64, 1024)
N, M = int(ln[:, 0].max()), int(ln[:, 1].max())
theta, A = datagen.theta_A(2, B, N, M)
t, a, lens = torch.from_numpy(theta).cuda(), torch.from_numpy(A).cuda(), torch.from_numpy(ln).cuda()
et = torch.ones(B, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
flag = 0x10000 if "nofill" in sys.argv else 0
import ctypes
fn = {}
for k, l in L.items():
    st = torch.empty(l.sdp_state_bytes(B, N, M) // 4, device="cuda")
    vt, E = torch.empty(B, device="cuda"), torch.empty(B, N, M, device="cuda")
    f = (lambda l, st, vt: lambda: l.sdp_forward_f32(t.data_ptr(), a.data_ptr(), st.data_ptr(), vt.data_ptr(), B, N, M, lens.data_ptr(), 0, 0, stream))(l, st, vt)
    g = (lambda l, st, E: lambda: l.sdp_backward_f32(et.data_ptr(), st.data_ptr(), E.data_ptr(), B, N, M, lens.data_ptr(), flag, 0, stream))(l, st, E)
    assert f() == 0 and g() == 0
    fn[k] = (f, g, vt, E)
    if k == "exp" and "noroute" in sys.argv:   # the shipped library once more without the second (thin-pair) launch: sdp_set_debug(32768)
        def wrap(h, l=l):
            def run():
                gpu_tune.set_debug(l, 32768); r = h(); gpu_tune.set_debug(l, 0); return r
            return run
        fn["exp-noroute"] = (wrap(f), wrap(g), vt, E)
torch.cuda.synchronize()
ref = fn["main"]
for k in fn:
    same = torch.equal(fn[k][2], ref[2]) and torch.equal(torch.nan_to_num(fn[k][3]), torch.nan_to_num(ref[3]))
    print(f"{k}: results {'equal to' if same else 'DIFFER from'} main")
res = {k: ([], []) for k in fn}
for rep in range(int(os.environ.get("REPS", 7))):
    for k, (f, g, _, _) in fn.items():
        for which, h in ((0, f), (1, g)):
            for _ in range(3): h()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(10): h()
            e.record(); torch.cuda.synchronize()
            res[k][which].append(s.elapsed_time(e) * 100)
for k, (rf, rb) in res.items():
    print(f"configs[2] lens {k:12s} fwd {np.median(rf):7.1f} us  bwd {np.median(rb):7.1f} us")
