import os, sys
sys.path.insert(0, "/root/repo/tools")
import gpu_tune
main = gpu_tune.load(os.path.join(gpu_tune.ROOT, "deepblast_amd", "libsdp_hip.so"))
for B in (96, 128, 160, 192, 224, 256, 320, 384):
    row = []
    for W in (4, 8):
        r = gpu_tune.run(main, B, 512, 512, (W, W, 0, 0), "fb")
        row.append(f"W={W}: fwd={r['fwd']:.0f} bwd={r['bwd']:.0f} seq={r['fwd;bwd']:.0f}")
    print(f"B={B}  " + "   ".join(row), flush=True)
