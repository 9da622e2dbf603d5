import os, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen, parity
from deepblast_amd._engine import get_engine
lib = get_engine().lib
for W in (1, 2, 3, 4, 8):
  for (B, N, M) in [(1,576,64),(1,640,64)]:
    for p in range(4): lib.sdp_set_waves(p, W)
    theta, A = datagen.theta_A(5, B, N, M)
    Z = datagen.normal(6, (B, N, M)); Et = np.ones(B, np.float32)
    ref = parity.oracle_all(theta, A, Et, Z, 0)
    got = parity.engine_all(theta, A, Et, Z, 0)
    e = parity.compare(got, ref)
    E, Er = got["E"], ref["E"]
    bad = np.argwhere(np.abs(E - Er) > 1e-3)
    print("W", W, B, N, M, {k: f"{v:.1e}" for k, v in e.items()}, "first bad E:", bad[:1].tolist(), "last bad:", bad[-1:].tolist(), "nbad", len(bad), flush=True)
