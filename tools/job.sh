#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
python - <<'PY'
import sys; sys.path.insert(0,'tools')
import gpu_tune, os
lib = gpu_tune.load(os.path.join(gpu_tune.ROOT, "deepblast_amd", "libsdp_hip.so"))
for B in (288, 320, 384, 448, 512, 640, 768, 1024):
    out=[]
    for W in (0, 2, 4):
        r = gpu_tune.run(lib, B, 512, 512, (W,W,0,0), "fb")
        out.append(f"W={W}: fwd={r['fwd']:.0f} bwd={r['bwd']:.0f} seq={r['fwd;bwd']:.0f}")
    print(f"B={B}: " + " | ".join(out), flush=True)
PY
