"""Tiny driver for counter collection: runs fwd (+bwd) a few times at batch size argv[1]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpu_tune
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
lib = gpu_tune.load(os.environ.get("SDP_LIB_PATH", os.path.join(gpu_tune.ROOT, "deepblast_amd", "libsdp_hip.so")))
r = gpu_tune.run(lib, B, N, N, (0, 0, 0, 0), "fb")
print(r)
