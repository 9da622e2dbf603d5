"""In-run A/B of library builds (main + build_variants/*): interleaved repetitions, median.  usage: ab.py [BxNxM ...] [W=n]"""
import glob, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpu_tune
libs = {"main": os.path.join(gpu_tune.ROOT, "deepblast_amd", "libsdp_hip.so")}
for p in sorted(glob.glob(os.path.join(gpu_tune.ROOT, "build_variants", "libsdp_*.so"))):
    libs[os.path.basename(p)[7:-3]] = p
L = {k: gpu_tune.load(v) for k, v in libs.items()}
shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:] if "x" in a] or [(16, 512, 512), (256, 512, 512)]
W = [int(a[2:]) for a in sys.argv[1:] if a.startswith("W=")]
W = W[0] if W else 0
WB = int(os.environ.get('WB', '0'))   # WB=n: force n waves in the BACKWARD sweep only
passes = "fba" if "adj" in sys.argv else "fb"  # adj: also the exact-state forward and the adjoint pair
for (B, N, M) in shapes:
    res = {k: [] for k in L}
    for rep in range(int(os.environ.get('REPS', '3'))):
        for k, l in L.items():
            res[k].append(gpu_tune.run(l, B, N, M, (W, WB or W, 0, 0), passes))
    for k in L:
        keys = res[k][0].keys()
        print(f"B={B} {N}x{M} W={W or 'auto'} {k:18s} " + " ".join(f"{kk}={np.median([r[kk] for r in res[k]]):.1f}" for kk in keys) + (f"  [fwd;bwd mean {np.mean([r['fwd;bwd'] for r in res[k]]):.1f} sd {np.std([r['fwd;bwd'] for r in res[k]]):.1f}]" if len(res[k]) > 3 else ""), flush=True)
