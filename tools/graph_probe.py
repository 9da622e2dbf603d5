"""Does a fwd+bwd step survive torch CUDA-graph capture (library launches go to the capturing stream)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen
from deepblast_amd import NeedlemanWunschDecoder
from deepblast_amd.distributed import ShardedAligner
B, N, M = 256, 512, 512
th, A = datagen.theta_A(1, B, N, M)
theta = torch.from_numpy(th).cuda(); a = torch.from_numpy(A).cuda()
al = ShardedAligner(NeedlemanWunschDecoder("softmax"), gather="none")
def step():
    return al.align(theta, a)["E_local"]
for _ in range(3): E0 = step()
torch.cuda.synchronize()
def timeit(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("eager ms/step:", timeit(step))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    Eg = step()
g.replay(); torch.cuda.synchronize()
print("graph ms/step:", timeit(g.replay), "equal:", torch.equal(Eg, E0))
