import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen
from deepblast_amd._engine import get_engine
eng = get_engine()
N, M = 512, 512
th, A = datagen.theta_A(1, 1, N, M)
t = torch.from_numpy(th).cuda(); a = torch.from_numpy(A).cuda()
def run(W):
    for p in range(4): eng.lib.sdp_set_waves(p, W)
    vt, st = eng.forward(t, a, 0)
    torch.cuda.synchronize()
    return st.cpu().numpy().view(np.uint32).copy(), float(vt[0])
ref, v1 = run(4)   # throughput build (K=32)
s, v = run(8)      # latency build (K=16)
tpad = 576
per = tpad * 64 * 3 // 2
d = np.nonzero(s != ref)[0]
print("Vt", v, v1, "differing words:", len(d), "of", len(s))
for x in d[:12]:
    strip, r = divmod(int(x), per); pair2, r2 = divmod(r, 192); lane, j = divmod(r2, 3)
    print("  strip", strip, "steps", 2 * pair2, "lane", lane, "word", j, hex(int(ref[x])), hex(int(s[x])), " col", 2 * pair2 - lane)
