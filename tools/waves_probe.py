"""fwd / bwd kernel time at the headline shape for every forced wave count (SDP_WAVES(w) travels with the call)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen
from deepblast_amd._engine import get_engine
B, N, M = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (256, 512, 512)))
eng = get_engine()
theta, A = datagen.theta_A(1, B, N, M)
t = torch.from_numpy(theta).cuda(); a = torch.from_numpy(A).cuda(); et = torch.ones(B, device="cuda")
def run(n):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(n):
        ev[0].record(); Vt, Q = eng.forward(t, a, 0); ev[1].record(); E = eng.backward(et, Q, (B, N, M), 0); ev[2].record()
        torch.cuda.synchronize(); tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
    return tf / n * 1e3, tb / n * 1e3
for rep in range(2):
    for w in (0, 3, 4, 5, 6, 7, 8):
        eng.force_waves = {} if w == 0 else {0: w, 1: w}
        run(3); f, b = run(15)
        print(f"W={w or 'auto'}: fwd {f:.1f} us  bwd {b:.1f} us  sum {f + b:.1f}", flush=True)
