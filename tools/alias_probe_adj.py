"""Upper bounds for the adjoint pair and the exact-state sweeps: time with tensors aliased to pair 0 (served from cache).
bit0 inputs, bit1 outputs, bit2 state; 7 = the instruction stream alone."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpu_tune
main = gpu_tune.load(os.path.join(gpu_tune.ROOT, "deepblast_amd", "libsdp_hip_exp.so"))
for mask in (0, 1, 2, 4, 7):
    gpu_tune.set_debug(main, mask)
    r = gpu_tune.run(main, 256, 512, 512, (0, 0, 0, 0), "fba")
    print(f"alias={mask}: " + " ".join(f"{k}={v:.1f}" for k, v in r.items()), flush=True)
gpu_tune.set_debug(main, 0)
