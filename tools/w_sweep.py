"""Latency / throughput of the automatic wave policy across batch sizes and shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gpu_tune  # noqa: E402

main = gpu_tune.load(os.path.join(gpu_tune.ROOT, "deepblast_amd", "libsdp_hip.so"))
for (B, N, M) in ((16, 512, 512), (64, 512, 512), (192, 512, 512), (256, 512, 512), (512, 512, 512), (32, 300, 350),
                  (64, 1024, 1024), (256, 1024, 1024), (64, 300, 2048)):
    r = gpu_tune.run(main, B, N, M, (0, 0, 0, 0), "fb")
    print(f"B={B} {N}x{M} auto: fwd={r['fwd']:.1f} bwd={r['bwd']:.1f} fwd;bwd={r['fwd;bwd']:.1f} "
          f"-> {2.0 * B * N * M / (r['fwd;bwd'] * 1e-6):.3e} cu/s", flush=True)
