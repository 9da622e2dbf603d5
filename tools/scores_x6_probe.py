"""The two scores kernels side by side: time, and error against a float64 einsum (the fp32 torch einsum beside them).
Needs the -DSDP_EXPERIMENTS build (sdp_set_debug(16) forces the f32-input MFMA kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch, torch.nn.functional as F
import gpu_tune
l = gpu_tune.load(os.path.join(gpu_tune.ROOT, "deepblast_amd", "libsdp_hip_exp.so"))
stream = torch.cuda.current_stream().cuda_stream
for (B, N, M, D, scale) in ((256, 512, 512, 512, 1.0), (8, 300, 200, 512, 1.0), (8, 300, 200, 512, 20.0), (4, 1000, 700, 1024, 4.0), (16, 128, 128, 64, 1.0)):
    g = torch.Generator(device="cuda").manual_seed(3)
    zx, zy, gx, gy = (torch.randn(B, n, D, device="cuda", generator=g) * (scale / D ** 0.5) for n in (N, M, N, M))
    th, A = torch.empty(B, N, M, device="cuda"), torch.empty(B, N, M, device="cuda")
    f = lambda: l.sdp_scores_f32(zx.data_ptr(), zy.data_ptr(), gx.data_ptr(), gy.data_ptr(), th.data_ptr(), A.data_ptr(), B, N, M, D, 0, stream)
    nb = min(B, 8)
    ref_t = F.softplus(torch.einsum("bid,bjd->bij", zx[:nb].double(), zy[:nb].double()))
    ref_a = F.logsigmoid(torch.einsum("bid,bjd->bij", gx[:nb].double(), gy[:nb].double()))
    t32 = F.softplus(torch.einsum("bid,bjd->bij", zx[:nb], zy[:nb]))
    out = {}
    for name, mask in (("bf16x6", 0), ("f32 mfma", 16)):
        l.sdp_set_debug(mask)
        assert f() == 0
        torch.cuda.synchronize()
        et, ea = float((th[:nb].double() - ref_t).abs().max()), float((A[:nb].double() - ref_a).abs().max())
        us = float(np.median([gpu_tune.timeit(f, 5) for _ in range(3)]))
        out[name] = (us, et, ea)
    l.sdp_set_debug(0)
    e32 = float((t32.double() - ref_t).abs().max())
    print(f"B={B} {N}x{M} D={D} scale {scale}: " + "  ".join(f"{k}: {v[0]:8.1f} us ({4.0 * B * N * M * D / v[0] / 1e6:6.1f} TF) err theta {v[1]:.2e} A {v[2]:.2e}" for k, v in out.items())
          + f"  | torch fp32 einsum err {e32:.2e}, max|theta| {float(ref_t.max()):.1f}", flush=True)
