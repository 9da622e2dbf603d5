"""Host-side cost of a call: forward + backward of a tiny problem through the autograd classes, through the engine, through
the C ABI, and replayed from a captured graph."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import datagen
from deepblast_amd import NeedlemanWunschDecoder
from deepblast_amd._engine import get_engine
eng = get_engine(); eng.init()
for (B, N, M) in ((1, 64, 64), (16, 128, 128), (16, 512, 512)):
    theta, A = datagen.theta_A(5, B, N, M)
    t, a = torch.from_numpy(theta).cuda(), torch.from_numpy(A).cuda()
    et = torch.ones(B, device="cuda")
    dec = NeedlemanWunschDecoder("softmax")
    def autograd_step():
        x = t.detach().requires_grad_(True)
        dec(x, a).sum().backward()
    def engine_step():
        Vt, Q = eng.forward(t, a, 0)
        return eng.backward(et, Q, (B, N, M), 0)
    def wall(f, n=200):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n): f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e6
    g = torch.cuda.CUDAGraph()
    engine_step(); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        out = engine_step()
    print(f"B={B} {N}x{M}: autograd step {wall(autograd_step):7.1f} us   engine fwd+bwd {wall(engine_step):7.1f} us   graph replay {wall(g.replay):7.1f} us", flush=True)
