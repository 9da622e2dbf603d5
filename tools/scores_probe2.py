"""wide (256 x 256) vs narrow (128 x 128) three-piece scores kernel: bit-identity and time; variants under build_variants/."""
import ctypes, glob, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import gpu_tune
B, N, M, D = 256, 512, 512, 512
for a in sys.argv[1:]:
    if "x" in a:
        B, N, M, D = (int(v) for v in a.split("x"))
zx, zy, gx, gy = (torch.randn(B, n, D, device="cuda") / D ** 0.5 for n in (N, M, N, M))
stream = torch.cuda.current_stream().cuda_stream
def run(l):
    th, A = torch.empty(B, N, M, device="cuda"), torch.empty(B, N, M, device="cuda")
    f = lambda: l.sdp_scores_f32(zx.data_ptr(), zy.data_ptr(), gx.data_ptr(), gy.data_ptr(), th.data_ptr(), A.data_ptr(), B, N, M, D, 0, stream)
    assert f() == 0
    return gpu_tune.timeit(f, 5), th, A
exp = gpu_tune.load(os.path.join(gpu_tune.ROOT, "deepblast_amd", "libsdp_hip_exp.so"))
gpu_tune.set_debug(exp, 0); tw, thw, Aw = run(exp)
gpu_tune.set_debug(exp, 32); tn, thn, An = run(exp)
gpu_tune.set_debug(exp, 0)
ref = torch.nn.functional.softplus(torch.einsum('bid,bjd->bij', zx.double(), zy.double())).float()
print(f"{B}x{N}x{M}x{D}: wide {tw:.1f} us, narrow {tn:.1f} us; bit-identical theta {torch.equal(thw, thn)} A {torch.equal(Aw, An)}; max err vs f64 {float((thw - ref).abs().max()):.2e}")
libs = {"main": os.path.join(gpu_tune.ROOT, "deepblast_amd", "libsdp_hip.so")}
for p in sorted(glob.glob(os.path.join(gpu_tune.ROOT, "build_variants", "libsdp_*.so"))):
    libs[os.path.basename(p)[7:-3]] = p
res = {}
loaded = {k: gpu_tune.load(path) for k, path in libs.items()}
names = list(loaded)
for rep in range(8):   # rotate the order: the first library of a round is measured at another clock than the last
    for k in names[rep % len(names):] + names[:rep % len(names)]:
        res.setdefault(k, []).append(run(loaded[k])[0])
for k, v in res.items():
    us = float(np.median(v))
    print(f"{k:14s} {us:9.1f} us  {4.0 * B * N * M * D / us / 1e6:7.1f} TFLOP/s fp32-equivalent, {6 * 4.0 * B * N * M * D / us / 1e6 / 2516.6:.3f} of the bf16 peak", flush=True)
