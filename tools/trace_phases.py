import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import gpu_tune, torch, numpy as np, datagen
lib = gpu_tune.load(os.path.join(gpu_tune.ROOT, "build_variants", "libsdp_trace.so"))
N = M = 512
names = ["issue loads", "poll+bcv", "lds reads in", "steps(math+st)", "publish", "flush", "write_block", "#windowed chunks"]
for B, W, dbg in ((256, 4, 7), (256, 4, 0), (16, 4, 0)):
    th, A = datagen.theta_A(1, min(B, 16), N, M)
    reps = (B + th.shape[0] - 1) // th.shape[0]
    t = torch.from_numpy(np.tile(th, (reps, 1, 1))[:B]).cuda(); a = torch.from_numpy(np.tile(A, (reps, 1, 1))[:B]).cuda()
    st = torch.empty(lib.sdp_state_bytes(B, N, M) // 4, device="cuda")
    vt = torch.zeros(B * 64, device="cuda")
    lib.sdp_set_waves(0, W)
    lib.sdp_set_waves(100, dbg)
    for _ in range(2):
        lib.sdp_forward_f32(t.data_ptr(), a.data_ptr(), st.data_ptr(), vt.data_ptr(), B, N, M, None, 0, 0, None)
    torch.cuda.synchronize()
    u = vt.cpu().numpy().reshape(B, 8, 8)
    lib.sdp_set_waves(100, 0)
    print(f"B={B} W={W} alias={dbg}: shader cycles per strip, mean over pairs; columns = strips 0..7")
    for i, nm in enumerate(names):
        print(f"  {nm:16s}", " ".join(f"{x:7.0f}" for x in u[:, :, i].mean(axis=0)))
    print(f"  {'total':16s}", " ".join(f"{x:7.0f}" for x in u[:, :, :7].sum(axis=2).mean(axis=0)))
for B, W in ((16, 8),):
    lib.sdp_set_waves(0, W)
    r = gpu_tune.run(lib, B, N, M, (W, 0, 0, 0), "fb")
    print("traced build wall time fwd (us):", r["fwd"])
