"""configs[2] lengths-aware: automatic policy vs forced 8 waves."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch, datagen
from gpu_configs import timeit, step
from deepblast_amd import NeedlemanWunschDecoder
from deepblast_amd._engine import get_engine
lib = get_engine().lib
B = 256
lens = datagen.lengths(2, B, 64, 1024)
N, M = int(lens[:, 0].max()), int(lens[:, 1].max())
th, A = datagen.theta_A(2, B, N, M)
th, A = torch.from_numpy(th).cuda(), torch.from_numpy(A).cuda()
ln = torch.from_numpy(lens).cuda()
dec = NeedlemanWunschDecoder("softmax")
work = int((lens[:, 0].astype(np.int64) * lens[:, 1]).sum())
for W in (0, 4, 6, 8):
    for p in range(2): lib.sdp_set_waves(p, W)
    ms = timeit(lambda: step(dec, th, A, ln), 5)
    msp = timeit(lambda: step(dec, th, A), 5)
    print(f"W={W or 'auto'}: lengths-aware {ms:.3f} ms ({2 * work / ms * 1e3:.3e} true cu/s)   padded {msp:.3f} ms")
