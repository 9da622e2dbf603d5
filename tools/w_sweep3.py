import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpu_tune
main = gpu_tune.load(os.path.join(gpu_tune.ROOT, "deepblast_amd", "libsdp_hip.so"))
for rnd in range(2):
    for (B, N, M) in ((16, 512, 512), (64, 512, 512), (256, 512, 512), (256, 1024, 1024)):
        r = gpu_tune.run(main, B, N, M, (0, 0, 0, 0), "fb")
        print(f"B={B} {N}x{M}: fwd={r['fwd']:.1f} bwd={r['bwd']:.1f} fwd;bwd={r['fwd;bwd']:.1f}", flush=True)
