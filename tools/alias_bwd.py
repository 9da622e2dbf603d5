"""Backward sweep under aliasing (bit1 outputs, bit2 state served from cache) for every experiments-capable library under build_variants/ + the exp build."""
import glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpu_tune
libs = {"exp": os.path.join(gpu_tune.ROOT, "deepblast_amd", "libsdp_hip_exp.so")}
for p in sorted(glob.glob(os.path.join(gpu_tune.ROOT, "build_variants", "libsdp_*.so"))):
    libs[os.path.basename(p)[7:-3]] = p
for name, path in libs.items():
    l = gpu_tune.load(path)
    out = []
    for mask in (0, 2, 4, 6):
        try:
            gpu_tune.set_debug(l, mask)
        except Exception:
            out.append("(no debug switch)")
            break
        r = gpu_tune.run(l, 256, 512, 512, (0, 0, 0, 0), "fb")
        out.append(f"alias={mask}: bwd={r['bwd']:.1f}")
    try:
        gpu_tune.set_debug(l, 0)
    except Exception:
        pass
    print(f"{name:10s} " + "  ".join(out), flush=True)
