"""Native against library backward of the scores over shapes (which is faster where)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import gpu_tune
from deepblast_amd import scores as sc
for (B, N, M, D) in [(16, 512, 512, 512), (64, 256, 256, 128), (256, 128, 128, 64), (8, 1000, 700, 1024), (32, 64, 64, 512), (128, 300, 200, 512), (4, 2000, 2000, 256), (256, 512, 512, 128)]:
    zx, zy, gx, gy = (torch.randn(B, n, D, device="cuda") / D ** 0.5 for n in (N, M, N, M))
    theta, A = sc.alignment_scores(zx, zy, gx, gy)
    g1, g2 = torch.randn_like(theta), torch.randn_like(A)
    tn = gpu_tune.timeit(lambda: sc._native_backward(zx, zy, gx, gy, theta, A, g1, g2), 5)
    tl = gpu_tune.timeit(lambda: sc._torch_backward(zx, zy, gx, gy, theta, A, g1, g2), 5)
    print(f"{B:4d} x {N:4d} x {M:4d} x {D:4d}: native {tn:8.1f} us   library {tl:8.1f} us   tiles256 {B * 2 * -(-N // 256) * -(-D // 256)}", flush=True)
