"""fwd/bwd time for every wave count at a few batch sizes (forced with sdp_set_waves)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpu_tune  # noqa: E402

main = gpu_tune.load(os.environ.get("SDP_LIB_PATH", os.path.join(gpu_tune.ROOT, "deepblast_amd", "libsdp_hip.so")))
shapes = [(16, 512, 512), (256, 512, 512), (512, 512, 512), (256, 1024, 1024)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
for (B, N, M) in shapes:
    for W in (2, 3, 4, 5, 6, 7, 8):
        r = gpu_tune.run(main, B, N, M, (W, W, 0, 0), "fb")
        print(f"B={B} {N}x{M} W={W}: fwd={r['fwd']:.1f} bwd={r['bwd']:.1f}", flush=True)
