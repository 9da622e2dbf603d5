"""Bit-for-bit comparison of the backward sweep (and the other sweeps) of library builds: main + build_variants/* against a
reference build (default: the first variant named on the command line, else `r4`).  usage: bitcmp.py [ref=name] [BxNxM ...] [sw] [x]
Every build runs ITS OWN forward sweep (the state formats may differ) and its backward sweep; E, Vt must be equal as bit patterns."""
import glob, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpu_tune
import torch
import datagen

libs = {"main": os.path.join(gpu_tune.ROOT, "deepblast_amd", "libsdp_hip.so")}
for p in sorted(glob.glob(os.path.join(gpu_tune.ROOT, "build_variants", "libsdp_*.so"))):
    libs[os.path.basename(p)[7:-3]] = p
ref = ([a[4:] for a in sys.argv[1:] if a.startswith("ref=")] or ["r4"])[0]
shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:] if a[0].isdigit()] or [(256, 512, 512), (64, 512, 512), (5, 130, 200), (16, 1024, 1024), (3, 700, 96), (300, 64, 64)]
L = {k: gpu_tune.load(v) for k, v in libs.items()}
bad = 0
for variant in ((0, 1) if "sw" in sys.argv else (0,)):
  for exact in ((0, 0x100) if "x" in sys.argv else (0,)):
    for (B, N, M) in shapes:
        th, A = datagen.theta_A(11, min(B, 64), N, M)
        reps = (B + th.shape[0] - 1) // th.shape[0]
        t = torch.from_numpy(np.tile(th, (reps, 1, 1))[:B] * np.float32(os.environ.get("THETA_SCALE", "1"))).cuda()
        a = torch.from_numpy(np.tile(A, (reps, 1, 1))[:B]).cuda()
        lens = None
        if "lens" in sys.argv:
            lens = torch.from_numpy(datagen.lengths(5, B, max(1, N // 8), N).clip(1, min(N, M))).cuda()
        et = torch.from_numpy(datagen.uniform(99, (B,)) + np.float32(0.5)).cuda()
        out = {}
        stream = torch.cuda.current_stream().cuda_stream
        for k, l in L.items():
            nb = max(l.sdp_state_bytes(B, N, M), l.sdp_state_d_bytes(B, N, M))
            st = torch.empty(nb // 4, device="cuda")
            vt = torch.empty(B, device="cuda")
            E = torch.full((B, N, M), 7.0, device="cuda")
            lp = None if lens is None else lens.data_ptr()
            assert l.sdp_forward_f32(t.data_ptr(), a.data_ptr(), st.data_ptr(), vt.data_ptr(), B, N, M, lp, variant | exact, 0, stream) == 0
            assert l.sdp_backward_f32(et.data_ptr(), st.data_ptr(), E.data_ptr(), B, N, M, lp, variant | exact, 0, stream) == 0
            torch.cuda.synchronize()
            out[k] = (vt.cpu().numpy().view(np.uint32), E.cpu().numpy().view(np.uint32))
        r = out[ref]
        for k in L:
            same = np.array_equal(out[k][0], r[0]) and np.array_equal(out[k][1], r[1])
            if not same:
                bad += 1
                d = np.abs(out[k][1].view(np.float32).astype(np.float64) - r[1].view(np.float32))
                print(f"MISMATCH {k} vs {ref}: variant={variant} exact={exact:#x} B={B} {N}x{M} lens={'y' if lens is not None else 'n'}: {int((out[k][1] != r[1]).sum())} words of E differ, max |d| {d.max():.3e}", flush=True)
        print(f"variant={variant} exact={exact:#x} B={B} {N}x{M}: {len(L)} builds compared against {ref}", flush=True)
print("bitcmp:", "ALL EQUAL" if bad == 0 else f"{bad} MISMATCHES")
