import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen, parity
B, N, M = 256, 512, 512
theta, A = datagen.theta_A(1, B, N, M)
got = parity.engine_all(theta, A, None, None, 0)
sel = [0, 100, 255]
alone = parity.engine_all(theta[sel], A[sel], None, None, 0)
print("Vt equal:", np.array_equal(alone["Vt"], got["Vt"][sel]))
d = np.argwhere(alone["E"] != got["E"][sel])
print("E mismatches:", len(d), d[:5].tolist(), d[-3:].tolist())
if len(d):
    i = tuple(d[0]); print(alone["E"][i], got["E"][sel][i])
    rows = np.unique(d[:, 1]); cols = np.unique(d[:, 2]); print("rows", rows[:10], rows[-5:], "cols", cols[:10], cols[-5:])
Et = np.full(3, 3.0, np.float32)
scaled = parity.engine_all(theta[sel], A[sel], Et, None, 0)
a, b = scaled["E"], 3.0 * alone["E"]
bad = np.argwhere(~np.isclose(a, b, rtol=1e-6, atol=1e-7))
print("linearity violations:", len(bad), bad[:3].tolist())
for i in bad[:5]:
    i = tuple(i); print(i, a[i], b[i], (a[i]-b[i])/b[i])
print("E[0,-1,-1]", got["E"][0, -1, -1])
again = parity.engine_all(theta[sel], A[sel], None, None, 0)
print("repeat equal:", np.array_equal(again["E"], alone["E"]))
