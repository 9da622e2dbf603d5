"""Build library variants under build_variants/.  usage: mkvariants.py name=-DFLAG[,-DFLAG2] ...  (name 'main' rebuilds the shipped
library).  Each variant is compiled by deepblast_amd/build.py (kernel groups and sources in parallel, one link), variants one after
the other."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepblast_amd import build


def mk(spec):
    name, _, flags = spec.partition("=")
    out = build.OUT if name == "main" else os.path.join(ROOT, "build_variants", f"libsdp_{name}.so")
    return build.build(True, extra=[f for f in flags.split(",") if f], out=out)


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "build_variants"), exist_ok=True)
    for spec in sys.argv[1:]:
        print(mk(spec), flush=True)
