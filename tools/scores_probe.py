"""Time sdp_scores_f32 for the main library and every build under build_variants/ (B=256, N=M=512, D=512)."""
import glob, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import gpu_tune
libs = {"main": os.path.join(gpu_tune.ROOT, "deepblast_amd", "libsdp_hip.so")}
for p in sorted(glob.glob(os.path.join(gpu_tune.ROOT, "build_variants", "libsdp_*.so"))):
    libs[os.path.basename(p)[7:-3]] = p
B, N, M, D = 256, 512, 512, 512
for a in sys.argv[1:]:
    if "x" in a:
        B, N, M, D = (int(v) for v in a.split("x"))
zx, zy, gx, gy = (torch.randn(B, n, D, device="cuda") / D ** 0.5 for n in (N, M, N, M))
th, A = torch.empty(B, N, M, device="cuda"), torch.empty(B, N, M, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
res = {}
for rep in range(3):
    for k, path in libs.items():
        l = gpu_tune.load(path)
        if not hasattr(l, "sdp_scores_f32"):
            continue
        f = lambda: l.sdp_scores_f32(zx.data_ptr(), zy.data_ptr(), gx.data_ptr(), gy.data_ptr(), th.data_ptr(), A.data_ptr(), B, N, M, D, 0, stream)
        assert f() == 0
        res.setdefault(k, []).append(gpu_tune.timeit(f, 5))
for k, v in res.items():
    us = float(np.median(v))
    print(f"{k:14s} {us:9.1f} us  {4.0 * B * N * M * D / us / 1e6:7.1f} TFLOP/s", flush=True)
