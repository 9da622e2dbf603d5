"""Steady-state A/B of library builds (main + build_variants/*): the fwd;bwd sequence looped without a pause, the GPU never idle
between the builds' turns.  tools/ab.py times bursts of 8 launches behind a synchronize -- a GPU that has just been idle runs the
same kernels ~5 % slower than one that has been busy for 20 ms (profiles/r05_step_gaps.txt), and what is bound by what shifts
with the clock.   usage: steady.py [BxNxM ...] [ROUNDS=n] [ITERS=n] [only=name,name]   (us per fwd;bwd, mean +- sd over rounds;
fwd / bwd: event pairs around every 16th iteration's sweeps)"""
import glob, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpu_tune
sys.path.insert(0, os.path.join(gpu_tune.ROOT, "tests"))
import datagen

libs = {"main": os.path.join(gpu_tune.ROOT, "deepblast_amd", "libsdp_hip.so")}
for p in sorted(glob.glob(os.path.join(gpu_tune.ROOT, "build_variants", "libsdp_*.so"))):
    libs[os.path.basename(p)[7:-3]] = p
arg = {a.split("=")[0]: a.split("=")[1] for a in sys.argv[1:] if "=" in a}
if arg.get("EXP"):   # EXP=1: the experiments build too (DBG=n: its sdp_set_debug mask -- 1 inputs, 2 outputs, 4 state aliased to pair 0)
    libs["exp"] = os.path.join(gpu_tune.ROOT, "deepblast_amd", "libsdp_hip_exp.so")
if "only" in arg:
    libs = {k: v for k, v in libs.items() if k in ("main", "exp") or k in arg["only"].split(",")}
L = {k: gpu_tune.load(v) for k, v in libs.items()}
if arg.get("DBG"):
    gpu_tune.set_debug(L["exp"], int(arg["DBG"], 0))
shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:] if "x" in a and "=" not in a] or [(256, 512, 512)]
ROUNDS, ITERS = int(arg.get("ROUNDS", 5)), int(arg.get("ITERS", 200))
WF, WB = int(arg.get("WF", 0)), int(arg.get("WB", 0))
FLAGS_B = int(arg.get("BFLAGS", "0"), 0)   # e.g. BFLAGS=0x800: SDP_NO_ZERO_SKIP
XS = 0x100 if (arg.get("X") or arg.get("ADJ")) else 0            # X=1: the exact (float2) state of the training path (SDP_EXACT_STATE)
ADJ = bool(arg.get("ADJ"))   # ADJ=1: the four sweeps of a training step (exact forward, exact backward, adjoint forward, adjoint backward)
for (B, N, M) in shapes:
    th, A = datagen.theta_A(1, min(B, 64), N, M)
    reps = (B + th.shape[0] - 1) // th.shape[0]
    t = torch.from_numpy(np.tile(th, (reps, 1, 1))[:B]).cuda()
    a = torch.from_numpy(np.tile(A, (reps, 1, 1))[:B]).cuda()
    vt, et, E = torch.empty(B, device="cuda"), torch.ones(B, device="cuda"), torch.empty(B, N, M, device="cuda")
    st = torch.empty(max((l.sdp_state_d_bytes(B, N, M) if XS else l.sdp_state_bytes(B, N, M)) for l in L.values()) // 4, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    if ADJ:
        std = torch.empty(max(l.sdp_state_d_bytes(B, N, M) for l in L.values()) // 4, device="cuda")
        Ed, z = torch.empty(B, N, M, device="cuda"), torch.randn(B, N, M, device="cuda")
    fn = {}
    for k, l in L.items():
        f = (lambda l: lambda: l.sdp_forward_f32(t.data_ptr(), a.data_ptr(), st.data_ptr(), vt.data_ptr(), B, N, M, None, ((WF & 0xf) << 12) | XS, 0, stream))(l)
        b = (lambda l: lambda: l.sdp_backward_f32(et.data_ptr(), st.data_ptr(), E.data_ptr(), B, N, M, None, ((WB & 0xf) << 12) | FLAGS_B | XS, 0, stream))(l)
        assert f() == 0 and b() == 0
        if ADJ:   # "b" becomes: exact backward, adjoint forward, adjoint backward
            b0 = b
            af = (lambda l: lambda: l.sdp_adjoint_forward_f32(st.data_ptr(), z.data_ptr(), None, vt.data_ptr(), std.data_ptr(), B, N, M, None, 0, 0, stream))(l)
            ab = (lambda l: lambda: l.sdp_adjoint_backward_f32(E.data_ptr(), st.data_ptr(), std.data_ptr(), Ed.data_ptr(), B, N, M, None, 0, 0, stream))(l)
            assert af() == 0 and ab() == 0
            b = (lambda b0, af, ab: lambda: (b0(), af(), ab()))(b0, af, ab)
        fn[k] = (f, b)
    res = {k: {"seq": [], "f": [], "b": []} for k in L}
    for k, (f, b) in fn.items():     # warm: 100 iterations each, back to back
        for _ in range(100):
            f(); b()
    for rnd in range(ROUNDS):
        order = list(fn) if rnd % 2 == 0 else list(fn)[::-1]
        pend = []
        for k in order:
            f, b = fn[k]
            for _ in range(20):
                f(); b()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(ITERS):
                f(); b()
            e.record()
            ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(12)]
            for e0, e1, e2 in ev:     # the split, behind the timed loop (event records cost the stream ~6 us each)
                for _ in range(4):
                    f(); b()
                e0.record(); f(); e1.record(); b(); e2.record()
            pend.append((k, s, e, ev))
        torch.cuda.synchronize()
        for k, s, e, ev in pend:
            res[k]["seq"].append(s.elapsed_time(e) / ITERS * 1e3)
            res[k]["f"].append(np.median([x.elapsed_time(y) for x, y, _ in ev]) * 1e3)
            res[k]["b"].append(np.median([y.elapsed_time(z) for _, y, z in ev]) * 1e3)
    for k in L:
        r = res[k]
        print(f"B={B} {N}x{M} {k:18s} fwd;bwd {np.mean(r['seq']):7.1f} +- {np.std(r['seq']):4.1f} us   (min {np.min(r['seq']):7.1f})   fwd {np.mean(r['f']):6.1f}  bwd {np.mean(r['b']):6.1f}", flush=True)
