"""Time of the batched device traceback beside the sweeps that produce its input (B pairs of N x M)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, datagen
from deepblast_amd._engine import get_engine
eng = get_engine()
for (B, N, M, ts) in ((256, 512, 512, 1.0), (256, 512, 512, 6.0), (64, 2048, 2048, 6.0), (1024, 128, 128, 6.0)):
    theta, A = datagen.theta_A(5, B, N, M)
    t = torch.from_numpy((theta * ts).astype(np.float32)).cuda(); a = torch.from_numpy(A).cuda()
    Vt, Q = eng.forward(t, a, 0)
    E = eng.backward(torch.ones(B, device="cuda"), Q, tuple(t.shape), 0)
    from deepblast_amd._engine import _ptr
    st, cn = eng.traceback(E)
    stream = torch.cuda.current_stream().cuda_stream
    call = lambda: eng.lib.sdp_traceback_i32(_ptr(E), _ptr(st), _ptr(cn), B, N, M, None, 0, stream)   # the raw entry point: no allocations in the loop
    for _ in range(3): call()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(20): call()
    ev[1].record(); torch.cuda.synchronize()
    print(f"B={B} {N}x{M} theta*{ts}: traceback {ev[0].elapsed_time(ev[1]) / 20 * 1e3:.0f} us, mean steps {float(cn.float().mean()):.0f}", flush=True)
