"""What the backward of the scores costs today (torch ops, scores.py::_Scores.backward) at B=256 N=M=D=512, piece by piece."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import gpu_tune
from deepblast_amd.scores import alignment_scores
B, N, M, D = 256, 512, 512, 512
zx, zy, gx, gy = (torch.randn(B, n, D, device="cuda") / D ** 0.5 for n in (N, M, N, M))
theta, A = alignment_scores(zx, zy, gx, gy)
g1, g2 = torch.randn_like(theta), torch.randn_like(A)
t = lambda f: gpu_tune.timeit(f, 5)
print(f"forward (sdp_scores_f32)            : {t(lambda: alignment_scores(zx, zy, gx, gy)):8.1f} us")
print(f"elementwise  g * (-expm1(-theta))   : {t(lambda: g1 * (-torch.expm1(-theta))):8.1f} us   (x2)")
ds = g1 * (-torch.expm1(-theta))
print(f"bmm(ds, zy)                         : {t(lambda: torch.bmm(ds, zy)):8.1f} us   (x2)")
print(f"bmm(ds^T, zx)                       : {t(lambda: torch.bmm(ds.transpose(1, 2), zx)):8.1f} us   (x2)")
zs = [z.clone().requires_grad_(True) for z in (zx, zy, gx, gy)]
def fb():
    th, a = alignment_scores(*zs)
    torch.autograd.backward([th, a], [g1, g2])
    for z in zs: z.grad = None
print(f"forward + backward through autograd : {t(fb):8.1f} us")
from deepblast_amd import scores as sc
nb = lambda: sc._native_backward(zx, zy, gx, gy, theta, A, g1, g2)
print(f"native backward (sdp_scores_backward_f32, both tensors): {t(nb):8.1f} us")
print(f"library backward (torch expm1 / mul / 4 x bmm)          : {t(lambda: sc._torch_backward(zx, zy, gx, gy, theta, A, g1, g2)):8.1f} us")
exp = gpu_tune.load(os.path.join(ROOT, "deepblast_amd", "libsdp_hip_exp.so"))
ws = torch.empty(exp.sdp_scores_backward_ws_bytes(B, N, M) // 4, device="cuda")
outs = [torch.empty_like(x) for x in (zx, zy, gx, gy)]
stream = torch.cuda.current_stream().cuda_stream
call = lambda: exp.sdp_scores_backward_f32(g1.data_ptr(), g2.data_ptr(), theta.data_ptr(), A.data_ptr(), zx.data_ptr(), zy.data_ptr(), gx.data_ptr(), gy.data_ptr(),
                                           ws.data_ptr(), *[o.data_ptr() for o in outs], B, N, M, D, 0, stream)
res = {}
for mask, name in ((0, "dS formed inside the dzy / dgy product"), (2048, "dS in a pass of its own")):
    gpu_tune.set_debug(exp, mask)
    assert call() == 0
    print(f"experiments build, {name:40s}: {t(call):8.1f} us")
    res[mask] = [o.clone() for o in outs]
gpu_tune.set_debug(exp, 0)
print("max |difference| between the two:", [float((x - y).abs().max()) for x, y in zip(res[0], res[2048])])
