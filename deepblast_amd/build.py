"""Build libsdp_hip.so for gfx950 with hipcc (in-tree, next to this file)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = [os.path.join(HERE, "csrc", "sdp_kernels.hip"), os.path.join(HERE, "csrc", "sdp_scores.hip"), os.path.join(HERE, "csrc", "sdp_ref.hip"),
       os.path.join(HERE, "csrc", "sdp_comm.hip"), os.path.join(HERE, "csrc", "sdp_api.hip")]
HDR = [os.path.join(HERE, "csrc", "sdp_kernels.h"), os.path.join(ROOT, "include", "sdp.h")]
OUT = os.path.join(HERE, "libsdp_hip.so")


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    return "hipcc"


def build(force=False, extra=(), out=None):
    """Compile the engine.  `extra`/`out` build experiment variants (e.g. -DSDP_K_FWD=32) next to it."""
    global OUT
    if out is not None:
        saved, OUT = OUT, out
        try:
            return build(True, extra)
        finally:
            OUT = saved
    newest = max(os.path.getmtime(f) for f in SRC + HDR)
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= newest:
        return OUT
    # -ffp-contract=off: every fused multiply-add in the kernels is written explicitly, so that the different
    # builds of a sweep (chunk length, masked / mask-free body) round identically and results do not depend on
    # which build or which body a cell happens to run in
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc"),
           *extra, *SRC, "-ldl", "-o", OUT]
    subprocess.check_call(cmd)
    return OUT


EXP_OUT = os.path.join(HERE, "libsdp_hip_exp.so")


def build_experiments(force=False):
    """The -DSDP_EXPERIMENTS build (sdp_set_debug: wrong-results timing switches, forced hand-off time-outs).
    Test and tuning infrastructure; the package never loads it."""
    newest = max(os.path.getmtime(f) for f in SRC + HDR)
    if not force and os.path.exists(EXP_OUT) and os.path.getmtime(EXP_OUT) >= newest:
        return EXP_OUT
    return build(True, extra=("-DSDP_EXPERIMENTS",), out=EXP_OUT) or EXP_OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
