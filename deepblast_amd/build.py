"""Build libsdp_hip.so for gfx950 with hipcc (in-tree, next to this file).

The sweep kernels (csrc/sdp_kernels.hip) are one template instantiated ~30 times; compiled as ONE translation unit that
takes two minutes.  The file therefore knows `-DSDP_GROUP=<g>` (one group of its kernels per translation unit), and the
groups, like the other sources, are compiled to objects in parallel and linked: ~35 s on 8 cores."""
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
KERNELS = os.path.join(HERE, "csrc", "sdp_kernels.hip")
KERNEL_GROUPS = 9   # SDP_GROUP = 0 .. 8 (sdp_kernels.hip, "SDP_IN_GROUP")
SRC = [KERNELS, os.path.join(HERE, "csrc", "sdp_scores.hip"), os.path.join(HERE, "csrc", "sdp_ref.hip"),
       os.path.join(HERE, "csrc", "sdp_comm.hip"), os.path.join(HERE, "csrc", "sdp_api.hip")]
HDR = [os.path.join(HERE, "csrc", "sdp_kernels.h"), os.path.join(ROOT, "include", "sdp.h")]
OUT = os.path.join(HERE, "libsdp_hip.so")


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.sep not in c or os.path.exists(c)):
            return c
    return "hipcc"


# -ffp-contract=off: every fused multiply-add in the kernels is written explicitly, so that the different
# builds of a sweep (chunk length, masked / mask-free body) round identically and results do not depend on
# which build or which body a cell happens to run in
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "csrc")]


def compile_and_link(out, extra=(), jobs=None):
    """Objects in parallel (the kernel file once per group), then one link."""
    units = [(KERNELS, [f"-DSDP_GROUP={g}"]) for g in range(KERNEL_GROUPS)] + [(s, []) for s in SRC if s != KERNELS]
    with tempfile.TemporaryDirectory(prefix="sdp_build_") as tmp:
        def cc(iu):
            i, (src, defs) = iu
            obj = os.path.join(tmp, f"u{i}.o")
            subprocess.check_call([hipcc(), *FLAGS, *extra, *defs, "-c", src, "-o", obj])
            return obj
        with ThreadPoolExecutor(jobs or min(len(units), os.cpu_count() or 4)) as ex:
            objs = list(ex.map(cc, enumerate(units)))
        subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-ldl", "-o", out])
    return out


def build(force=False, extra=(), out=None):
    """Compile the engine.  `extra`/`out` build experiment variants (e.g. -DSDP_K_FWD=32) next to it."""
    out = out or OUT
    newest = max(os.path.getmtime(f) for f in SRC + HDR)
    if not force and os.path.exists(out) and os.path.getmtime(out) >= newest:
        return out
    return compile_and_link(out, tuple(extra))


EXP_OUT = os.path.join(HERE, "libsdp_hip_exp.so")


def build_experiments(force=False):
    """The -DSDP_EXPERIMENTS build (sdp_set_debug: wrong-results timing switches, forced hand-off time-outs).
    Test and tuning infrastructure; the package never loads it."""
    newest = max(os.path.getmtime(f) for f in SRC + HDR)
    if not force and os.path.exists(EXP_OUT) and os.path.getmtime(EXP_OUT) >= newest:
        return EXP_OUT
    return build(True, extra=("-DSDP_EXPERIMENTS",), out=EXP_OUT)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
