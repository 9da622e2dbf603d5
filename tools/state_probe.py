import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen
from deepblast_amd._engine import get_engine
eng = get_engine()
N, M = 640, 64
th, A = datagen.theta_A(5, 1, N, M)
t = torch.from_numpy(th).cuda(); a = torch.from_numpy(A).cuda()
def run(W):
    for p in range(4): eng.lib.sdp_set_waves(p, W)
    vt, st = eng.forward(t, a, 0)
    e = eng.backward(torch.ones(1, device="cuda"), st, (1, N, M), 0) if False else None
    torch.cuda.synchronize()
    return st.cpu().numpy().view(np.uint32).copy(), float(vt[0])
ref, v1 = run(1)
tpad = 128; per = tpad * 64
for W in (8, 8, 8, 4):
    s, v = run(W)
    d = np.nonzero(s != ref)[0]
    for x in d[:8]:
        print("   word", int(x), hex(int(ref[x])), hex(int(s[x])))
    print("W", W, "Vt", v, v1, "differing words:", len(d), [(int(x) // per, (int(x) % per) // 256, (int(x) % 256) // 4, int(x) % 4) for x in d[:6]], "(strip, t/4, lane, j)")
