"""Where the adjoint sweeps' time goes, from shader-cycle stamps inside the kernels (experiments build, sdp_set_trace):
per 16-step chunk [top, outputs of the previous chunk flushed, boundary values there + rows requested, steps done,
published] of pairs 0 and 128, all four waves, both strips of each wave, at B=256 N=M=512.
usage: [ALIAS=7] adj_trace.py [b|f] [B N M]      (b: the adjoint backward sweep, f: the adjoint forward sweep;
ALIAS: 1 inputs, 2 outputs, 4 states of every pair aliased to pair 0's, i.e. served from cache -- timing only)"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import gpu_tune
lib = gpu_tune.load(os.environ.get("SDP_TRACE_LIB") or os.path.join(ROOT, "deepblast_amd", "libsdp_hip_exp.so"))
lib.sdp_set_trace.restype, lib.sdp_set_trace.argtypes = ctypes.c_int, [ctypes.c_void_p]
which = sys.argv[1] if len(sys.argv) > 1 else "b"
B, N, M = 256, 512, 512
if len(sys.argv) > 4:
    B, N, M = (int(v) for v in sys.argv[2:5])
bit, key = (8192, "abwd") if which == "b" else (16384, "afwd")
trace = torch.zeros(4 * 4 * 4 * 40 * 8, dtype=torch.int64, device="cuda")
alias = int(os.environ.get("ALIAS", "0"))
gpu_tune.set_debug(lib, bit | alias)
r0 = gpu_tune.run(lib, B, N, M, (0, 0, 0, 0), "fba")
lib.sdp_set_trace(trace.data_ptr())
r = gpu_tune.run(lib, B, N, M, (0, 0, 0, 0), "fba")
lib.sdp_set_trace(None); gpu_tune.set_debug(lib, 0)
print(f"alias={alias} {key}: {r0[key]:.1f} us untraced, {r[key]:.1f} us traced   (all: " + "  ".join(f"{k} {v:.1f}" for k, v in r0.items() if isinstance(v, float)) + ")")
t = trace.cpu().numpy().reshape(4, 4, 4, 40, 8)
for pair in ((0,) if B < 129 else (0, 2)):
    starts = t[pair][..., 0]
    t0 = starts[starts > 0].min()
    print(f"pair {64 * pair}: per wave and strip round: first chunk start .. last chunk end (cycles); mean cycles per chunk that ran its steps: "
          "flush | boundary wait + requests | steps | publish | whole; chunks skipped as zero")
    for w in range(4):
        for rd in range(4):
            x = t[pair, w, rd]
            nb = int((x[:, 0] > 0).sum())
            if nb < 6:
                continue
            x = x[:nb]
            if x[-1, 4] == 0:
                x, nb = x[:-1], nb - 1
            if which == "f":   # the forward-running sweep stores its state inside the steps: no flush phase, no stamp 1
                x = x.copy(); x[:, 1] = x[:, 0]
            flush, acq, steps, pub = x[:, 1] - x[:, 0], x[:, 2] - x[:, 1], x[:, 3] - x[:, 2], x[:, 4] - x[:, 3]
            total = np.append(x[1:, 0] - x[:-1, 0], 0)
            ran = steps >= 200
            ran[:2] = False; ran[-2:] = False
            if os.environ.get("TRACE_TIMELINE"):
                print(f"    wave {w} round {rd} chunk starts: " + " ".join(str(int(u - t0)) for u in x[:, 0]) + f" | end {int(x[-1, 4] - t0)}")
            if os.environ.get("TRACE_BLOCKS"):
                for name, v in (("flush", flush), ("acq", acq), ("steps", steps), ("pub", pub), ("total", total)):
                    print(f"    wave {w} round {rd} per chunk: {name:6s}" + " ".join(str(int(u)) for u in v))
            m = lambda v: v[ran].mean() if ran.any() else 0.0
            print(f"  wave {w} round {rd}: {x[0, 0] - t0:8d} .. {x[-1, 4] - t0:8d}  chunks {nb} ({int((steps < 200).sum())} zero)  flush {m(flush):6.0f}  acquire {m(acq):6.0f}  "
                  f"steps {m(steps):6.0f}  publish {m(pub):5.0f}  whole chunk {m(total):6.0f}")
