"""Phase timings inside sdp_scores_x6w_kernel (build_variants/libsdp_trace.so, -DSDP_XW_TRACE=1,-DSDP_XW_ABL=4): cycle stamps
of waves 0 and 4 of two workgroups at the phase boundaries of slabs 8..15."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import gpu_tune
B, N, M, D = 256, 512, 512, 512
l = gpu_tune.load(os.path.join(gpu_tune.ROOT, "build_trace", "libsdp_trace.so"))
zx, zy, gx, gy = (torch.randn(B, n, D, device="cuda") / D ** 0.5 for n in (N, M, N, M))
th, A = torch.zeros(B + 1, N, M, device="cuda"), torch.zeros(B, N, M, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    assert l.sdp_scores_f32(zx.data_ptr(), zy.data_ptr(), gx.data_ptr(), gy.data_ptr(), th.data_ptr(), A.data_ptr(), B, N, M, D, 0, stream) == 0
torch.cuda.synchronize()
t = th[B].view(-1)[:2 * 2 * 8 * 5 * 2].cpu().numpy().view(np.uint64).reshape(2, 2, 8, 5).astype(np.int64)
for wg in range(2):
    for g in range(2):
        x = t[wg, g]
        print(f"workgroup {wg} group {g}: per slab [mem phase, wait at barrier, matrix phase, wait at barrier] cycles; period")
        for i in range(8):
            mem, w1, mat, w2 = x[i, 1] - x[i, 0], x[i, 2] - x[i, 1], x[i, 3] - x[i, 2], x[i, 4] - x[i, 3]
            per = x[i + 1, 0] - x[i, 0] if i < 7 else 0
            print(f"   slab {8 + i}: {mem:6d} {w1:6d} {mat:6d} {w2:6d}   period {per}")

nwg = 2048
m = th[B].view(-1)[2048:2048 + nwg * 2 * 4 * 2].cpu().numpy().view(np.uint64).reshape(nwg, 2, 4).astype(np.int64)[:, 0]
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record(); l.sdp_scores_f32(zx.data_ptr(), zy.data_ptr(), gx.data_ptr(), gy.data_ptr(), th.data_ptr(), A.data_ptr(), B, N, M, D, 0, stream); e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
print(f"kernel {ms * 1e3:.0f} us; mean prologue {(m[:, 1] - m[:, 0]).mean():.0f}, loop {(m[:, 2] - m[:, 1]).mean():.0f}, epilogue {(m[:, 3] - m[:, 2]).mean():.0f} units")
for xcd in range(8):   # the cycle counters of the eight XCDs are not synchronised: one analysis per XCD (workgroup h runs on XCD h % 8)
    mx = m[xcd::8]
    span = mx[:, 3].max() - mx[:, 0].min()
    dur = mx[:, 3] - mx[:, 0]
    starts = np.sort(mx[:, 0] - mx[:, 0].min())
    print(f"XCD {xcd}: span {span} units -> {ms * 1e6 / span:.3f} ns per unit; concurrency {dur.sum() / span:.1f} of 32 CUs; start deciles {[int(starts[int(q * (len(starts) - 1) / 8)]) for q in range(9)]}")
