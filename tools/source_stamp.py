#!/usr/bin/env python
"""Identity of the kernel sources a measurement belongs to: sha256 over deepblast_amd/csrc/* and include/sdp.h, plus
the launch plan (kernel build ids) of the headline configuration.  bench.py compares it with profiles/traffic.json."""
import ctypes
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["deepblast_amd/csrc/sdp_kernels.hip", "deepblast_amd/csrc/sdp_kernels.h", "deepblast_amd/csrc/sdp_api.hip", "deepblast_amd/csrc/sdp_comm.hip",
         "deepblast_amd/csrc/sdp_ref.hip", "include/sdp.h"]


def source_sha():
    h = hashlib.sha256()
    for f in FILES:
        with open(os.path.join(ROOT, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    extra = os.path.join(ROOT, "deepblast_amd/csrc/sdp_scores.hip")
    if os.path.exists(extra):
        with open(extra, "rb") as fh:
            h.update(b"deepblast_amd/csrc/sdp_scores.hip\0" + fh.read())
    return h.hexdigest()


def plan_ids(B=256, N=512, M=512, cus=256):
    sys.path.insert(0, ROOT)
    from deepblast_amd import _lib
    lib = _lib.load()
    names = {0: "sdp_fwd_kernel", 1: "sdp_bwd_kernel", 2: "sdp_adj_fwd_kernel", 3: "sdp_adj_bwd_kernel", 4: "sdp_bwd_lat_kernel",
             5: "sdp_fwd_x_kernel", 6: "sdp_fwd_lat_kernel", 7: "sdp_bwd_x_kernel", 8: "sdp_bwd_x_lat_kernel", 9: "sdp_fwd_x_tp_kernel",
             36: "sdp_bwd_pipe_kernel", 37: "sdp_fwd_c_kernel", 38: "sdp_fwd_x_tp_c_kernel", 39: "sdp_fwd_lat_c_kernel", 40: "sdp_fwd_x_c_kernel"}
    out = {}
    for label, pass_, exact in (("fwd", 0, 0), ("bwd", 1, 0), ("fwd_exact", 0, 1), ("bwd_exact", 1, 1), ("adj_fwd", 2, 0), ("adj_bwd", 3, 0)):
        kid, chunk, waves, lds = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
        lib.sdp_plan(pass_, B, N, M, 0, exact, cus, ctypes.byref(kid), ctypes.byref(chunk), ctypes.byref(waves), ctypes.byref(lds))
        out[label] = {"kernel": names.get(kid.value, str(kid.value)), "chunk": chunk.value, "waves": waves.value, "lds_bytes": lds.value}
    return out


if __name__ == "__main__":
    git = None
    try:
        import subprocess
        git = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
    except Exception:
        pass
    print(json.dumps({"source_sha256": source_sha(), "git_head": git, "plan_B256_512x512": plan_ids()}, indent=1))
