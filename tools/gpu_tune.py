#!/usr/bin/env python
"""Timing experiments on the GPU box: variant libraries x waves x batch sizes.
usage: python tools/gpu_tune.py [lib=path ...]  (default: main lib + everything under build_variants/)"""
import ctypes
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import datagen  # noqa: E402
from deepblast_amd import _lib  # noqa: E402


def load(path):
    lib = ctypes.CDLL(path)
    for name, (res, args) in _lib.SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:  # an older build kept for comparison
            continue
        fn.restype, fn.argtypes = res, args
    return lib


def set_debug(lib, mask):
    """Experiment switches exist only in -DSDP_EXPERIMENTS builds (deepblast_amd/libsdp_hip_exp.so)."""
    if hasattr(lib, "sdp_set_debug"):
        lib.sdp_set_debug.restype, lib.sdp_set_debug.argtypes = ctypes.c_int, [ctypes.c_int]
        return lib.sdp_set_debug(mask)
    if mask:
        raise RuntimeError("this library has no sdp_set_debug: load deepblast_amd/libsdp_hip_exp.so")
    return 0


def timeit(fn, n=8):
    fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us


def run(lib, B, N, M, waves=(0, 0, 0, 0), passes="fb"):
    th, A = datagen.theta_A(1, min(B, 64), N, M)
    th = th * np.float32(os.environ.get("THETA_SCALE", "1"))   # (steeper scores: more of E underflows to zero)
    reps = (B + th.shape[0] - 1) // th.shape[0]
    t = torch.from_numpy(np.tile(th, (reps, 1, 1))[:B]).cuda()
    a = torch.from_numpy(np.tile(A, (reps, 1, 1))[:B]).cuda()
    st = torch.empty(lib.sdp_state_bytes(B, N, M) // 4, device="cuda")
    dbytes = lib.sdp_state_d_bytes(B, N, M) if hasattr(lib, "sdp_state_d_bytes") else lib.sdp_state_bytes(B, N, M)
    std = torch.empty(dbytes // 4, device="cuda") if "a" in passes else None
    vt = torch.empty(B, device="cuda")
    et = torch.ones(B, device="cuda")
    E = torch.empty(B, N, M, device="cuda")
    Ed = torch.empty(B, N, M, device="cuda") if "a" in passes else None
    z = torch.randn(B, N, M, device="cuda") if "a" in passes else None
    stream = torch.cuda.current_stream().cuda_stream
    wf = [(w & 0xf) << 12 for w in waves]   # include/sdp.h SDP_WAVES(w): travels with the call
    out = {}
    f = lambda: lib.sdp_forward_f32(t.data_ptr(), a.data_ptr(), st.data_ptr(), vt.data_ptr(), B, N, M, None, wf[0], 0, stream)
    b = lambda: lib.sdp_backward_f32(et.data_ptr(), st.data_ptr(), E.data_ptr(), B, N, M, None, wf[1], 0, stream)
    assert f() == 0
    out["fwd"] = timeit(f)
    out["bwd"] = timeit(b)
    out["fwd;bwd"] = timeit(lambda: (f(), b()))
    # the two sweeps INSIDE the back-to-back sequence (events between them): the backward sweep behind a forward sweep is not the
    # backward sweep re-run on a state that has been lying in memory (what "bwd" above times)
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(8)]
    f(); b()
    torch.cuda.synchronize()
    for e0, e1, e2 in evs:
        e0.record(); f(); e1.record(); b(); e2.record()
    torch.cuda.synchronize()
    out["seq_f"] = float(np.median([e0.elapsed_time(e1) for e0, e1, _ in evs])) * 1e3
    out["seq_b"] = float(np.median([e1.elapsed_time(e2) for _, e1, e2 in evs])) * 1e3
    if "a" in passes:
        stx = torch.empty(dbytes // 4, device="cuda")  # exact state for the adjoint sweeps
        if hasattr(lib, "sdp_state_d_bytes"):
            assert lib.sdp_forward_f32(t.data_ptr(), a.data_ptr(), stx.data_ptr(), vt.data_ptr(), B, N, M, None, 0x100 | wf[0], 0, stream) == 0
            out["fwd_x"] = timeit(lambda: lib.sdp_forward_f32(t.data_ptr(), a.data_ptr(), stx.data_ptr(), vt.data_ptr(), B, N, M, None, 0x100 | wf[0], 0, stream))
        else:
            stx = st
        af = lambda: lib.sdp_adjoint_forward_f32(stx.data_ptr(), z.data_ptr(), None, vt.data_ptr(), std.data_ptr(), B, N, M, None, wf[2], 0, stream)
        ab = lambda: lib.sdp_adjoint_backward_f32(E.data_ptr(), stx.data_ptr(), std.data_ptr(), Ed.data_ptr(), B, N, M, None, wf[3], 0, stream)
        out["afwd"] = timeit(af)
        out["abwd"] = timeit(ab)
        fx = lambda: lib.sdp_forward_f32(t.data_ptr(), a.data_ptr(), stx.data_ptr(), vt.data_ptr(), B, N, M, None, 0x100 | wf[0], 0, stream)
        bx = lambda: lib.sdp_backward_f32(et.data_ptr(), stx.data_ptr(), E.data_ptr(), B, N, M, None, 0x100 | wf[1], 0, stream)
        out["bwd_x"] = timeit(bx)
        out["train4"] = timeit(lambda: (fx(), bx(), af(), ab()))   # the four sweeps of a training step, back to back
    return out


def main():
    libs = {"main": os.path.join(ROOT, "deepblast_amd", "libsdp_hip.so")}
    for p in sorted(glob.glob(os.path.join(ROOT, "build_variants", "libsdp_*.so"))):
        libs[os.path.basename(p)[7:-3]] = p
    only = [a for a in sys.argv[1:] if not a.startswith("-")]
    print("device:", torch.cuda.get_device_name(0))
    main_lib = load(libs["main"])
    print("== waves x batch (main lib), N=M=512, us")
    for B in (64, 256, 512, 1024):
        for W in (1, 2, 3, 4):
            if "--wb" not in sys.argv and B != 256:
                continue
            r = run(main_lib, B, 512, 512, (W, W, W, W), "fba" if B == 256 else "fb")
            print(f"B={B:5d} W={W}: " + "  ".join(f"{k}={v:8.1f}" for k, v in r.items()), flush=True)
    exp_lib = load(os.path.join(ROOT, "deepblast_amd", "libsdp_hip_exp.so"))
    print("== aliasing experiments (experiments build, B=256, W=4): dbg bit0 inputs, bit1 outputs, bit2 state alias pair 0")
    for dbg in (0, 1, 2, 4, 5, 6, 7):
        set_debug(exp_lib, dbg)
        r = run(exp_lib, 256, 512, 512, (0, 0, 0, 0), "fb")
        print(f"dbg={dbg}: " + "  ".join(f"{k}={v:8.1f}" for k, v in r.items()), flush=True)
    set_debug(exp_lib, 0)
    print("== variants at B=256 W=4, us (3 interleaved rounds, min)")
    sel = {n: load(pth) for n, pth in libs.items()
           if not ((only and n not in only) or (not only and "--variants" not in sys.argv and n != "main"))}
    best = {}
    for rnd in range(3):
        for name, l in sel.items():
            r = run(l, 256, 512, 512, (0, 0, 0, 0), "fba" if "--adj" in sys.argv else "fb")
            for k, v in r.items():
                best.setdefault(name, {})[k] = min(v, best.get(name, {}).get(k, 1e9))
    for name, r in best.items():
        print(f"{name:10s}: " + "  ".join(f"{k}={v:8.1f}" for k, v in r.items()), flush=True)
    print("== shapes (main lib, W auto), us and cell-updates/s")
    shapes = ((256, 512, 512), (256, 512, 500), (256, 512, 520), (256, 512, 528), (256, 512, 544), (256, 512, 576),
              (256, 1024, 1024), (256, 1024, 1040), (256, 128, 128), (2048, 64, 64))
    if "--pitch" in sys.argv:
        for name in ("abl13", "abl14", "abl12"):
            if name in libs:
                l = load(libs[name])
                for (B, N, M) in ((256, 512, 512), (256, 512, 528)):
                    r = run(l, B, N, M)
                    print(f"{name} B={B} N={N} M={M}: fwd={r['fwd']:.1f} bwd={r['bwd']:.1f}", flush=True)
    for (B, N, M) in shapes:
        r = run(main_lib, B, N, M)
        cu = 2.0 * B * N * M / ((r["fwd"] + r["bwd"]) * 1e-6)
        print(f"B={B} N={N} M={M}: fwd={r['fwd']:.1f} bwd={r['bwd']:.1f}  {cu:.3e} cu/s", flush=True)


if __name__ == "__main__":
    main()
