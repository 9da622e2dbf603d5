"""Upper bounds: time with some tensors aliased to pair 0 (traffic served from cache).  bit0 inputs, bit1 outputs, bit2 state."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpu_tune
main = gpu_tune.load(os.environ.get("SDP_LIB_PATH", os.path.join(gpu_tune.ROOT, "deepblast_amd", "libsdp_hip_exp.so")))
for B in (256,):
    for mask in (0, 1, 2, 4, 5, 6, 7):
        gpu_tune.set_debug(main, mask)
        r = gpu_tune.run(main, B, 512, 512, (0, 0, 0, 0), "fb")
        print(f"B={B} alias={mask}: fwd={r['fwd']:.1f} bwd={r['bwd']:.1f}", flush=True)
gpu_tune.set_debug(main, 0)
