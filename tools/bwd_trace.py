"""Where the backward sweep's time goes, from shader-cycle stamps inside the kernel (experiments build, sdp_set_trace):
per 32-step chunk [top, outputs of the previous chunk flushed, boundary values there, steps done, published] of pairs 0
and 128, all four waves, both strips of each wave, at B=256 N=M=512.  usage: bwd_trace.py [alias mask] [B N M]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import gpu_tune
lib = gpu_tune.load(os.environ.get("SDP_TRACE_LIB") or os.path.join(ROOT, "deepblast_amd", "libsdp_hip_exp.so"))   # (SDP_TRACE_LIB: another -DSDP_EXPERIMENTS build)
lib.sdp_set_trace.restype, lib.sdp_set_trace.argtypes = ctypes.c_int, [ctypes.c_void_p]
B, N, M = 256, 512, 512
alias = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if len(sys.argv) > 4:
    B, N, M = (int(v) for v in sys.argv[2:5])
trace = torch.zeros(4 * 4 * 4 * 40 * 8, dtype=torch.int64, device="cuda")
gpu_tune.set_debug(lib, alias | 1024)   # 1024: the stamps of the backward sweep, not the forward's
r0 = gpu_tune.run(lib, B, N, M, (0, 0, 0, 0), "fb")
lib.sdp_set_trace(trace.data_ptr())
r = gpu_tune.run(lib, B, N, M, (0, 0, 0, 0), "fb")
lib.sdp_set_trace(None); gpu_tune.set_debug(lib, 0)
print(f"alias={alias}: bwd {r0['bwd']:.1f} us untraced, {r['bwd']:.1f} us traced")
t = trace.cpu().numpy().reshape(4, 4, 4, 40, 8)
for pair in ((0,) if B < 129 else (0, 2)):
    t0 = t[pair][t[pair] > 0].min()
    print(f"pair {64 * pair}: per wave and strip round: first chunk start .. last chunk end (cycles); mean cycles per interior chunk: flush | boundary wait + read | steps | publish | rest (input reads, loop)")
    for w in range(4):
        for rd in range(4):
            x = t[pair, w, rd]
            nb = int((x[:, 0] > 0).sum())
            if nb < 6:
                continue
            x = x[:nb]
            if x[-1, 4] == 0:   # the extra iteration that only flushes the last chunk's outputs
                x, nb = x[:-1], nb - 1
            flush, acq, steps, pub = x[:, 1] - x[:, 0], x[:, 2] - x[:, 1], x[:, 3] - x[:, 2], x[:, 4] - x[:, 3]
            total = x[1:, 0] - x[:-1, 0]
            mid = slice(3, nb - 3)
            if os.environ.get("TRACE_TIMELINE"):   # absolute times: when every chunk of the strip began, and when its last one was published
                print(f"    wave {w} round {rd} chunk starts: " + " ".join(str(int(u - t0)) for u in x[:, 0]) + f" | end {int(x[-1, 4] - t0)}")
            if os.environ.get("TRACE_BLOCKS"):
                for name, v in (("flush", flush), ("acq", acq), ("steps", steps), ("pub", pub), ("total", total)):
                    print(f"    wave {w} round {rd} per chunk: {name:6s}" + " ".join(str(int(u)) for u in v))
            print(f"  wave {w} round {rd}: {x[0, 0] - t0:8d} .. {x[-1, 4] - t0:8d}  chunks {nb}  flush {flush[mid].mean():6.0f}  acquire {acq[mid].mean():6.0f}  steps {steps[mid].mean():6.0f}  "
                  f"publish {pub[mid].mean():5.0f}  whole chunk {total[mid].mean():6.0f}")
