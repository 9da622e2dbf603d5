"""Ablation timing at small and full batch (latency-bound vs bandwidth-bound regimes)."""
import glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpu_tune
libs = {"main": os.path.join(gpu_tune.ROOT, "deepblast_amd", "libsdp_hip.so")}
for p in sorted(glob.glob(os.path.join(gpu_tune.ROOT, "build_variants", "libsdp_*.so"))):
    libs[os.path.basename(p)[7:-3]] = p
L = {k: gpu_tune.load(v) for k, v in libs.items()}
for (B, W) in ((16, 8), (16, 4), (256, 8), (256, 4)):
    for name, l in L.items():
        r = gpu_tune.run(l, B, 512, 512, (W, W, 0, 0), "fb")
        print(f"B={B} W={W} {name:16s} fwd={r['fwd']:.1f} bwd={r['bwd']:.1f}", flush=True)
