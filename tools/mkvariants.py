"""Build library variants under build_variants/ in parallel.  usage: mkvariants.py name=-DFLAG[,-DFLAG2] ...  (name 'main' rebuilds the shipped library)"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepblast_amd import build


def mk(spec):
    name, _, flags = spec.partition("=")
    out = build.OUT if name == "main" else os.path.join(ROOT, "build_variants", f"libsdp_{name}.so")
    cmd = [build.hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "deepblast_amd", "csrc"),
           *[f for f in flags.split(",") if f], *build.SRC, "-ldl", "-o", out]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "build_variants"), exist_ok=True)
    with ThreadPoolExecutor(8) as ex:
        for o in ex.map(mk, sys.argv[1:]):
            print(o)
