"""Time decomposition of the sweeps: every library under build_variants/ (built with -DSDP_EXPERIMENTS and some
-DSDP_ABL=mask, see sdp_kernels.hip) at B=256 512x512, with real traffic and with all pairs aliased to pair 0
(cache-served).  usage: python tools/abl_probe.py [BxNxM]"""
import glob, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpu_tune
libs = {"exp": os.path.join(gpu_tune.ROOT, "deepblast_amd", "libsdp_hip_exp.so")}
for p in sorted(glob.glob(os.path.join(gpu_tune.ROOT, "build_variants", "libsdp_*.so"))):
    libs[os.path.basename(p)[7:-3]] = p
shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:] if "x" in a] or [(256, 512, 512)]
L = {k: gpu_tune.load(v) for k, v in libs.items()}
for (B, N, M) in shapes:
    for alias in (0, 7):
        res = {k: [] for k in L}
        for rep in range(3):
            for k, l in L.items():
                try:
                    gpu_tune.set_debug(l, alias)
                except RuntimeError:
                    if alias:
                        continue
                res[k].append(gpu_tune.run(l, B, N, M, (0, 0, 0, 0), "fb"))
                gpu_tune.set_debug(l, 0) if hasattr(l, "sdp_set_debug") else None
        for k in L:
            if res[k]:
                print(f"B={B} {N}x{M} alias={alias} {k:14s} " + " ".join(f"{kk}={np.median([r[kk] for r in res[k]]):.1f}" for kk in res[k][0]), flush=True)
