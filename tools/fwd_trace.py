"""Where the forward sweep's time goes, from shader-cycle stamps inside the kernel (experiments build, sdp_set_trace):
per 16-step block [start, inputs + hand-off ready, recurrence done, published] of pairs 0, 64, 128, 192, all four waves,
both strips of each wave, at B=256 N=M=512."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import gpu_tune
lib = gpu_tune.load(os.environ.get("SDP_TRACE_LIB") or os.path.join(ROOT, "deepblast_amd", "libsdp_hip_exp.so"))
lib.sdp_set_trace.restype, lib.sdp_set_trace.argtypes = ctypes.c_int, [ctypes.c_void_p]
B, N, M = 256, 512, 512
alias = int(sys.argv[1]) if len(sys.argv) > 1 else 0
if len(sys.argv) > 4:
    B, N, M = (int(v) for v in sys.argv[2:5])
trace = torch.zeros(4 * 4 * 4 * 40 * 8, dtype=torch.int64, device="cuda")
gpu_tune.set_debug(lib, alias)
r0 = gpu_tune.run(lib, B, N, M, (0, 0, 0, 0), "fb")
lib.sdp_set_trace(trace.data_ptr())
r = gpu_tune.run(lib, B, N, M, (0, 0, 0, 0), "fb")
lib.sdp_set_trace(None); gpu_tune.set_debug(lib, 0)
print(f"alias={alias}: fwd {r0['fwd']:.1f} us untraced, {r['fwd']:.1f} us traced")
t = trace.cpu().numpy().reshape(4, 4, 4, 40, 8)
for pair in ((0,) if B < 129 else (0, 2)):
    t0 = t[pair][t[pair] > 0].min()
    print(f"pair {64 * pair}: per wave and strip round: first block start, last block end (cycles since the pair's first stamp); mean cycles per block: wait | compute | publish | gap to next block")
    for w in range(4):
        for rd in range(4):
            x = t[pair, w, rd]
            nb = int((x[:, 0] > 0).sum())
            if nb < 10:
                continue
            x = x[:nb]
            wait, comp, pub = x[:, 1] - x[:, 0], x[:, 2] - x[:, 1], x[:, 3] - x[:, 2]
            gap = x[1:, 0] - x[:-1, 3]
            mid = slice(6, nb - 2)   # interior blocks
            odd = x[7:nb - 2:2]
            if os.environ.get("TRACE_CHUNK"):
                ev, od = x[6:nb - 2:2], x[7:nb - 2:2]
                print(f"     chunk pipeline: load_block issue {(ev[:, 5] - ev[:, 4]).mean():.0f}  (odd block end -> write_block start {(od[:, 4] - od[:, 3]).mean():.0f})  write_block {(od[:, 5] - od[:, 4]).mean():.0f}  "
                      f"write_block end -> next load_block start {(x[8:nb - 2:2, 4] - x[7:nb - 3:2, 5]).mean():.0f}  load_block end -> block start {(ev[:, 0] - ev[:, 5]).mean():.0f}")
            if False:
                print(f"     odd blocks: start->check {(odd[:, 4] - odd[:, 0]).mean():.0f}  write_block {(odd[:, 5] - odd[:, 4]).mean():.0f}  load_block {(odd[:, 6] - odd[:, 5]).mean():.0f}  prefetch issue {(odd[:, 7] - odd[:, 6]).mean():.0f}  ->compute {(odd[:, 1] - odd[:, 7]).mean():.0f};  even blocks wait {(x[6:nb - 2:2, 1] - x[6:nb - 2:2, 0]).mean():.0f}")
            if os.environ.get("TRACE_TIMELINE"):
                print(f"    wave {w} round {rd} block starts: " + " ".join(str(int(u - t0)) for u in x[:, 0]) + f" | end {int(x[-1, 3] - t0)}")
            if os.environ.get("TRACE_BLOCKS"):
                print(f"    wave {w} round {rd} per block: wait " + " ".join(str(int(v)) for v in wait))
                print(f"    wave {w} round {rd} per block: comp " + " ".join(str(int(v)) for v in comp))
                print(f"    wave {w} round {rd} per block: publ " + " ".join(str(int(v)) for v in pub))
                print(f"    wave {w} round {rd} per block: gap  " + " ".join(str(int(v)) for v in gap))
            print(f"  wave {w} round/part {rd}: {x[0, 0] - t0:8d} .. {x[-1, 3] - t0:8d}  blocks {nb}  "
                  f"wait {wait[mid].mean():6.0f} compute {comp[mid].mean():6.0f} publish {pub[mid].mean():5.0f} gap(even->odd block, odd->next chunk) {gap[6:-2:2].mean():6.0f} {gap[7:-2:2].mean():6.0f}")
