"""Where do the microseconds between the kernels of a step go?  usage (GPU box):
     rocprofv3 --kernel-trace -d gpurun_out/gap/<idiom> -o t -- python tools/gap_probe.py run <idiom>
     python tools/gap_probe.py report gpurun_out/gap/<idiom>
idioms: sum_backward (the metric's literal idiom), grad (torch.autograd.grad with a ones cotangent), engine (the C ABI
calls with nothing between them), sum_backward_keepgrad (theta.grad is not reset: PyTorch accumulates with an add kernel)."""
import csv, glob, os, sys
import numpy as np


def run(idiom, steps=60):
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import datagen
    from deepblast_amd import NeedlemanWunschDecoder
    from deepblast_amd._engine import get_engine
    B, N, M = 256, 512, 512
    theta, A = datagen.theta_A(1, B, N, M)
    t = torch.from_numpy(theta).cuda().requires_grad_()
    a = torch.from_numpy(A).cuda()
    dec = NeedlemanWunschDecoder("softmax")
    eng = get_engine()
    ones = torch.ones(B, device="cuda")
    for it in range(steps):
        if idiom == "sum_backward":
            t.grad = None
            dec(t, a).sum().backward()
        elif idiom == "sum_backward_keepgrad":
            dec(t, a).sum().backward()
        elif idiom == "grad":
            torch.autograd.grad(dec(t, a), t, ones)
        elif idiom == "engine":
            Vt, Q = eng.forward(t.detach(), a, 0)
            eng.backward(Q, a, ones, 0)
        else:
            raise SystemExit(f"unknown idiom {idiom}")
    torch.cuda.synchronize()


def report(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[len(rows) // 3:]   # (past the warm-up)
    gaps, durs = {}, {}
    for p, r in zip(rows, rows[1:]):
        n = r["Kernel_Name"][:34]
        gaps.setdefault(n, []).append((int(r["Start_Timestamp"]) - int(p["End_Timestamp"])) / 1e3)
        durs.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    fw = [int(r["Start_Timestamp"]) for r in rows if r["Kernel_Name"].startswith("sdp_fwd")]
    print(f"{os.path.basename(d.rstrip('/'))}: step (fwd start to fwd start) median {np.median(np.diff(fw)) / 1e3:.1f} us")
    for n in gaps:
        print(f"    {n:36s} n={len(gaps[n]):3d}  duration median {np.median(durs[n]):7.1f} us   gap before it median {np.median(gaps[n]):5.1f} us (mean {np.mean(gaps[n]):.1f})")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        for d in sys.argv[2:]:
            report(d)
