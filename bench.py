#!/usr/bin/env python
"""bench.py -- headline benchmark: NW soft-DP forward+backward, B=256 N=M=512 per GPU.

Metric (BASELINE.json): DP cell-updates/s, one cell-update = one recurrence evaluation at one
(b,i,j); a step `Vt = dec(theta, A); Vt.sum().backward()` on (B,N,M) is 2*B*N*M cell-updates.
Inputs are synthetic (theta ~ U[0,1), A = -U[0,1), fp32, tests/datagen.py) and resident in HBM
before the timed region.  N>1: one process per GPU (torch.distributed.run), every rank aligns its
own B pairs (weak scaling, no data-path collective) and the step ends with the RCCL all-gather
that collects the terminal scores Vt from all ranks (--gather e also gathers E).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel of the mode (the one with the longest
mean launch; sdp_fwd_kernel in the headline mode): algorithmic bytes (12 B per cell-update for the forward
and backward sweeps, 32 B for each adjoint sweep; flops on the matrix pipe when the scores
GEMM dominates) over its mean launch duration measured with HIP events on the launch stream, in a loop of the
same step that runs right BEHIND the timed region (round 6: the timed region itself carries no instrument at
any --steps -- an event record costs the stream ~6 us, tools/gap_probe.py).  `cpu_baseline` is the CPU oracle (a port of
deepblast/nw.py, oracle/sdp_oracle.c) timed on this box's host cores, rank 0 at N=1 only.
"""
import argparse
import contextlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ALGO_BYTES_PER_CELL_UPDATE = 12  # theta 4 + A 4 + state 4 | state 4 + A 4 + E 4


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # (0.06 s of GPU time at the headline shape; see --warmup)
    ap.add_argument("--warmup", type=int, default=50,
                    help="untimed steps before the timed region.  The defaults measure the steady state: K = 20 behind W = 3 -- 7 ms "
                         "after an idle GPU -- reads 0.316-0.321 ms per step where K = 200 behind W = 50 reads 0.289-0.297 on the same "
                         "box, interleaved (clocks still rising, and the fixed costs of a short region; DESIGN.md 4)")
    ap.add_argument("--B", type=int, default=256, help="pairs per GPU")
    ap.add_argument("--N", type=int, default=512)
    ap.add_argument("--M", type=int, default=512)
    ap.add_argument("--variant", choices=["nw", "sw"], default="nw")
    ap.add_argument("--mode", choices=["fwdbwd", "align+traceback", "train", "scores+dp", "train-mce", "train-mce-fused"], default="fwdbwd",
                    help="fwdbwd: headline; train: decode -> loss -> backward (adds the adjoint pair); scores+dp: theta/A "
                         "from (B,N,D) embeddings on the matrix cores (alignment.py:122-123), then the headline step; "
                         "train-mce: decode -> MatrixCrossEntropy (the reference's training loss, trainer.py:154-171) -> backward; "
                         "train-mce-fused: the same as one op, the loss gradient seeding the adjoint sweep inside the kernel")
    ap.add_argument("--D", type=int, default=512, help="embedding width of --mode scores+dp (reference default n_embed)")
    ap.add_argument("--gather", choices=["vt", "e", "paths", "none"], default="vt")
    ap.add_argument("--e-chunks", type=int, default=1, help="pieces of the backward sweep / E gather when E is gathered (1 = one collective after the sweep)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pairs", type=int, default=256, help="pairs in the CPU-baseline sample")
    return ap.parse_args()


class KernelTimer:
    """Brackets engine launches with a pair of events on the launch stream -- in the kernel-time loop BEHIND the timed
    region (`kernel_loop`), never inside it.

    Every `every`-th launch of each kernel is bracketed (default: every fourth one): a pair of event records costs
    the step ~3 us per kernel (measured: 0.402 vs 0.387 ms per step with all launches bracketed / none), so the loop
    they ride in is not the one the headline is read from."""

    def __init__(self, every=4):
        self.spans = []
        self.enabled = False
        self.every = every
        self.seen = {}

    @contextlib.contextmanager
    def __call__(self, name):
        if self.enabled:
            self.seen[name] = self.seen.get(name, 0) + 1
        if not self.enabled or self.seen[name] % self.every != 0:
            yield
            return
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        s.record()
        try:
            yield
        finally:
            e.record()
            self.spans.append((name, s, e))

    def means_ms(self):
        acc = {}
        for name, s, e in self.spans:
            acc.setdefault(name, []).append(s.elapsed_time(e))
        return {k: float(np.mean(v)) for k, v in acc.items()}


class ClockSampler:
    """Shader clock and socket power of the device WHILE the step loops (sysfs of the device's PCI function, polled from a
    thread; `rocm-smi` as the fallback).  The same kernel ran 171-210 us on the boxes of round 4: a bench line without the
    clock it was measured at cannot be compared with another round's."""

    def __init__(self, dev_index):
        self.dir = None
        self.samples = []
        self._stop = False
        try:
            pr = torch.cuda.get_device_properties(dev_index)
            want = f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            import glob
            for d in glob.glob("/sys/class/drm/card*/device"):
                if os.path.basename(os.path.realpath(d)) == want and os.path.exists(os.path.join(d, "pp_dpm_sclk")):
                    self.dir = d
                    break
        except Exception:   # noqa: BLE001
            self.dir = None

    def _read(self):
        out = {}
        try:
            for name, key in (("pp_dpm_sclk", "sclk_mhz"), ("pp_dpm_mclk", "mclk_mhz"), ("pp_dpm_fclk", "fclk_mhz")):
                with open(os.path.join(self.dir, name)) as f:
                    cur = [ln for ln in f.read().splitlines() if ln.rstrip().endswith("*")]
                if cur:
                    out[key] = float(cur[0].split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
            import glob
            for pw in glob.glob(os.path.join(self.dir, "hwmon", "hwmon*", "power1_*")):
                if pw.endswith(("power1_input", "power1_average")):
                    with open(pw) as f:
                        out["power_w"] = float(f.read().strip()) / 1e6
                    break
        except (OSError, ValueError, IndexError):
            pass
        return out

    def _smi(self):
        import subprocess
        try:
            txt = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
            j = json.loads(txt[txt.index("{"):])
            card = next(iter(j.values()))
            out = {}
            for k, v in card.items():
                kl = k.lower()
                if "sclk" in kl and "clock speed" in kl:
                    out["sclk_mhz"] = float(str(v).strip("()").lower().replace("mhz", ""))
                elif "mclk" in kl and "clock speed" in kl:
                    out["mclk_mhz"] = float(str(v).strip("()").lower().replace("mhz", ""))
                elif "power" in kl and "(w)" in kl:
                    out["power_w"] = float(v)
            return out
        except Exception:   # noqa: BLE001
            return {}

    def run(self, seconds, busy, rounds):
        """busy(): enqueue some steps (returns quickly); called exactly `rounds` times (the same number on every rank: the
        step may hold a collective).  Samples while the device works for about `seconds`."""
        import threading

        def poll():
            while not self._stop:
                smp = self._read() if self.dir else self._smi()
                if smp:
                    self.samples.append(smp)
                time.sleep(0.02)
        th = threading.Thread(target=poll, daemon=True)
        t0 = time.perf_counter()
        th.start()
        for _ in range(rounds):
            busy()
            torch.cuda.synchronize()
        seconds = time.perf_counter() - t0
        self._stop = True
        th.join(timeout=30)
        smp = self.samples[len(self.samples) // 4:] or self.samples   # (the first quarter: clocks still ramping)
        if not smp:
            return {"sclk_mhz": None, "power_w": None, "samples": 0, "how": "no sysfs access and no rocm-smi on this box"}
        out = {k: float(np.median([x[k] for x in smp if k in x])) for k in ("sclk_mhz", "mclk_mhz", "fclk_mhz", "power_w") if any(k in x for x in smp)}
        out["samples"] = len(smp)
        out["how"] = (("sysfs pp_dpm_* / hwmon of the device" if self.dir else "rocm-smi --showclocks --showpower") +
                      f", median over ~{seconds:.1f} s of the same step looping right after the timed region")
        return out


def cpu_baseline(args):
    """Time the CPU oracle (port of deepblast/nw.py) on a bounded sample of the same workload."""
    import datagen
    from oracle import oracle
    oracle.build()
    variant = 0 if args.variant == "nw" else 1
    Bc = min(args.cpu_pairs, args.B)
    theta, A = datagen.theta_A(1, Bc, args.N, args.M)
    et = np.ones(Bc, np.float32)
    reps = 0
    t0 = time.perf_counter()
    while True:  # faithful: the reference loops `for b in range(B)` serially around single-threaded code
        _, Q = oracle.forward(theta, A, variant, omp=False)
        oracle.backward(et, Q, variant, omp=False)
        reps += 1
        if time.perf_counter() - t0 > 10.0 or reps >= 3:
            break
    dt = (time.perf_counter() - t0) / reps
    out = {"value": 2.0 * Bc * args.N * args.M / dt, "unit": "cell-updates/s", "cores": 1, "kind": "port",
           "sample": f"oracle/sdp_oracle.c (port of deepblast/nw.py, -O2, f64 internals) fwd+bwd on {Bc} of the "
                     f"{args.B} pairs ({args.N}x{args.M}), {reps} rep(s), 1 thread"}
    ncpu = os.cpu_count() or 1
    t0 = time.perf_counter()
    _, Q = oracle.forward(theta, A, variant, omp=True)
    oracle.backward(et, Q, variant, omp=True)
    dt = time.perf_counter() - t0
    out["all_cores_value"] = 2.0 * Bc * args.N * args.M / dt
    out["all_cores"] = ncpu
    try:
        with open("/proc/cpuinfo") as f:
            models = [ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")]
        out["cpu_model"] = models[0] if models else "unknown"
    except OSError:
        out["cpu_model"] = "unknown"
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: re-run this script under torch.distributed.run with one
    rank per GPU of this node (RCCL over xGMI), rendezvous on 127.0.0.1.  Fails loudly if the node has fewer
    than N devices -- a run that silently timed fewer GPUs would be a wrong number."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if have < args.gpus:
        raise SystemExit(f"[bench] --gpus {args.gpus} requested but only {have} device(s) visible")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    # RCCL exchanges buffer handles between the ranks of a node through HIP IPC, and the host driver of this fleet only
    # supports the dmabuf kind: with the legacy mode ncclCommInit / the first collective fails with
    # "hipIpcGetMemHandle: invalid argument".  Must be set before the first device call; a caller's own setting wins.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    assert torch.cuda.is_available(), "bench.py needs a ROCm device (no CPU fallback)"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return spawn_ranks(args)   # `python bench.py --gpus N`: create the N ranks ourselves
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"[bench] WORLD_SIZE={world} does not match --gpus {args.gpus}: refusing to time a "
                         f"different job than the one asked for")
    # test hook (tests/test_distributed_nccl_gpu.py): exercise the N > 1 flow of this script on a 1-GPU box -- all ranks
    # on device 0, collectives over gloo.  Never set for a measurement: the JSON line says so in config.backend.
    # test hook: BENCH_FORCE_DIST=1 takes the N > 1 flow (process group, gather, fences, secondary measurements) with a world
    # of ONE rank -- on the nccl backend that sends every collective of the flow through RCCL on a 1-GPU box
    multi = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    shared = os.environ.get("BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if shared:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"[bench] rank {rank}: LOCAL_RANK={local_rank} but only {torch.cuda.device_count()} device(s) visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import datagen
    from deepblast_amd import NeedlemanWunschDecoder, SmithWatermanDecoder
    from deepblast_amd._engine import get_engine
    from deepblast_amd.distributed import ShardedAligner

    B, N, M = args.B, args.N, args.M
    theta_np, A_np = datagen.theta_A(1 + rank, B, N, M)  # BASELINE.md config C2 (seed 1) per rank
    theta = torch.from_numpy(theta_np).to(dev)
    A = torch.from_numpy(A_np).to(dev)
    Zl = torch.from_numpy(datagen.normal(7, (B, N, M))).to(dev) if args.mode == "train" else None
    dec = (NeedlemanWunschDecoder if args.variant == "nw" else SmithWatermanDecoder)("softmax")
    # the headline step is the metric's own idiom, literally: Vt = dec(theta, A); Vt.sum().backward(); the
    # same sweeps with the cotangent handed to torch.autograd.grad directly are reported next to it (`direct_cotangent`)
    aligner = ShardedAligner(dec, gather=args.gather if multi else "none", e_chunks=args.e_chunks,
                             idiom="sum_backward" if args.mode == "fwdbwd" else "grad")
    eng = get_engine()
    # kernel launch times: a loop of KSTEPS steps behind the timed region, every fourth launch of each kernel bracketed
    KSTEPS = max(16, min(args.steps, 64))
    timer = KernelTimer(every=4)
    eng.launch_hook = None

    emb = None
    if args.mode == "scores+dp":
        from deepblast_amd.scores import alignment_scores
        sc = 2.0 / np.sqrt(args.D)
        emb = [torch.from_numpy((datagen.normal(20 + i, (B, n, args.D)) * sc).astype(np.float32)).to(dev)
               for i, n in enumerate((N, M, N, M))]

    mce = None
    if args.mode.startswith("train-mce"):
        from deepblast_amd.losses import MatrixCrossEntropy, decode_loss
        mce = MatrixCrossEntropy()
        Yt = torch.from_numpy((datagen.uniform(30, (B, N, M)) < 0.05).astype(np.float32)).to(dev)
        Gm = torch.from_numpy((datagen.uniform(31, (B, N, M)) < 0.8).astype(np.float32)).to(dev)
        xl, yl = [N] * B, [M] * B

    walk = {"stream": None}

    def step():
        if mce is not None:
            t = theta.detach().requires_grad_(True)
            a = A.detach().requires_grad_(True)
            if args.mode == "train-mce-fused":
                loss, _ = decode_loss(dec, mce, t, a, Yt, xl, yl, Gm)
            else:
                loss = mce(Yt, dec.decode(t, a), xl, yl, Gm)
            loss.backward()
            return t.grad
        if args.mode == "scores+dp":
            th, ga = alignment_scores(*emb)     # one MFMA launch: both GEMMs + softplus / logsigmoid
            return aligner.align(th, ga)["E_local"]
        if args.mode == "fwdbwd":
            out = aligner.align(theta, A)      # Vt = dec(theta, A); Vt.sum().backward(); all-gather Vt under the backward sweep
            return out["E_local"]
        if args.mode == "align+traceback":     # inference: the alignment matrix and its arg-max walk (alignment.py:165-170, batched)
            out = aligner.align(theta, A)
            if out.get("paths") is not None:
                return out["paths"]
            E = out["E_local"]
            ws = walk["stream"]
            if ws is None:
                return eng.traceback(E)
            # (secondary figure `pipelined_walk`: the walk of this batch on a second stream, beside the sweeps of the next batch)
            ws.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(ws):
                paths = eng.traceback(E)
            E.record_stream(ws)   # the allocator must not hand E's memory to the next step before the walk has read it
            return paths
        t = theta.detach().requires_grad_(True)
        a = A.detach().requires_grad_(True)    # decode() differentiates w.r.t. (theta, A) like the reference
        aln = dec.decode(t, a)                 # forward + backward kernels (create_graph)
        # synthetic loss <aln, Z> as one dot product (forward: one reduction; backward: one scaling of Z) instead of
        # mul + sum (three elementwise kernels): the framework's share of the step, not the sweeps'
        torch.dot(aln.reshape(-1), Zl.reshape(-1)).backward()   # adjoint forward + adjoint backward kernels
        return t.grad

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    def timed(nsteps):
        """EXACTLY nsteps steps between two fences; -> (wall seconds, max over ranks; ms per step between two events).

        ONE pair of events around the whole loop, not one per step: an event record between two dependent kernels costs the
        stream ~6 us (rocprofv3 kernel trace of this loop with a record after every step: a 6.0-6.3 us hole in front of every
        forward sweep, none without; tools/gap_probe.py) -- 2 % of a 0.32 ms step spent on the measurement itself."""
        m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fence()
        t0 = time.perf_counter()
        m0.record()
        for i in range(nsteps):
            step()
        m1.record()
        fence()
        dt = time.perf_counter() - t0
        if multi:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, m0.elapsed_time(m1) / nsteps

    elapsed, ms_per_step_events = timed(args.steps)   # the headline: nothing but the steps between the two fences

    def kernel_loop(tm):
        """per-kernel mean launch times: KSTEPS more steps of the same loop, right behind the timed region (the GPU never idles in
        between beyond the fence), with an event pair around every fourth launch of each kernel"""
        eng.launch_hook = tm
        tm.enabled = True
        try:
            for _ in range(KSTEPS):
                step()
            fence()
        finally:
            tm.enabled = False
            eng.launch_hook = None
        return tm.means_ms()
    cells = B * N * M if args.variant == "nw" else B * (N - 1) * (M - 1)
    per_step_updates = (4 if args.mode.startswith("train") else 2) * cells
    value = world * per_step_updates * args.steps / elapsed
    ms = kernel_loop(timer)

    no_skip = clocks = pipelined = None

    def emit(e_gather, e_gather_one, paths_gather, direct=None):
        """rank 0: the ONE JSON line (called once: after the secondary measurements, or by the watchdog below)."""
        if rank == 0:
            # dominant kernel of THIS mode = the sweep / GEMM with the longest mean launch; its algorithmic bytes per cell
            # follow SURVEY.md 8(d): fwd and bwd 12 B, adjoint fwd (a3) and adjoint bwd (a4) 32 B each (reference layout)
            def algo_bytes_per_cell(name):
                return 32 if name.startswith("sdp_adj_") else ALGO_BYTES_PER_CELL_UPDATE
            cand = {k: v for k, v in ms.items() if k.startswith(("sdp_fwd", "sdp_bwd", "sdp_adj_", "sdp_scores"))}
            dom = max(cand, key=cand.get) if cand else "sdp_fwd_kernel"
            dom_ms = ms.get(dom, float("nan"))
            dom_bytes = cells * algo_bytes_per_cell(dom)
            achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
            # HBM bytes per launch from the PMC passes (profiles/traffic.json, tools/gpu_round.sh): only if that file was
            # measured on exactly these kernel sources -- a stale figure is reported as null, not as a number
            traffic, traffic_stamp, tj = None, None, {}
            tf = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tf) and (B, N, M, args.variant) == (256, 512, 512, "nw"):
                try:
                    sys.path.insert(0, os.path.join(ROOT, "tools"))
                    import source_stamp
                    tj = json.load(open(tf))
                    traffic_stamp = tj.get("_stamp", {}).get("source_sha256")
                    if traffic_stamp == source_stamp.source_sha():
                        traffic = tj.get(dom, {}).get("hbm_bytes_per_launch")
                    else:
                        tj = {}
                        print("[bench] profiles/traffic.json was measured on other kernel sources: roofline.traffic = null", file=sys.stderr)
                except (OSError, ValueError, ImportError):
                    traffic, tj = None, {}
            # every sweep of the step, each with BOTH fractions: `frac` on the algorithmic bytes (what the
            # roofline is graded on) and `real_frac` on the bytes the kernel actually moved (PMC traffic / launch time) -- a
            # kernel that moves fewer bytes than the algorithmic figure (the backward sweep skips exact-zero chunks) must not
            # show a flattering `frac` without the honest one next to it
            per_kernel = {}
            for k, v_ms in cand.items():
                if k.startswith("sdp_scores"):
                    continue
                ab = cells * algo_bytes_per_cell(k)
                tr = tj.get(k, {}).get("hbm_bytes_per_launch") if tj else None
                per_kernel[k] = {"launch_ms": v_ms, "algorithmic_bytes_per_launch": ab, "frac": ab / (v_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                 "traffic": tr, "real_frac": (tr / (v_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr else None}
            step_algo_bytes = cells * sum(algo_bytes_per_cell(k) for k in per_kernel) if per_kernel else per_step_updates * ALGO_BYTES_PER_CELL_UPDATE
            line = {
                "metric": {"fwdbwd": "DP cell-updates/sec (fwd+bwd)", "align+traceback": "DP cell-updates/sec (fwd+bwd + batched traceback)", "train": "DP cell-updates/sec (train: fwd+bwd+adjoint pair)",
                           "scores+dp": "DP cell-updates/sec (scores from embeddings + fwd+bwd)",
                           "train-mce": "DP cell-updates/sec (train: decode + MatrixCrossEntropy + backward)",
                           "train-mce-fused": "DP cell-updates/sec (train: fused decode + MatrixCrossEntropy + backward)"}[args.mode],
                "value": value, "unit": "cell-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": elapsed / args.steps * 1e3, "ms_per_step_events": ms_per_step_events,
                "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"{args.variant.upper()} soft-DP " + {"fwdbwd": "fwd+bwd", "align+traceback": "fwd+bwd+traceback", "train": "decode+loss.backward",
                                                                            "scores+dp": f"scores(D={args.D})+fwd+bwd", "train-mce": "decode+MatrixCrossEntropy.backward",
                                                                            "train-mce-fused": "fused decode+MatrixCrossEntropy.backward"}[args.mode] +
                                       f", B={B} per GPU, N={N}, M={M}, random theta/A "
                                       + ("(BASELINE.json configs[1])" if world == 1 else
                                          f"(BASELINE.json configs[4] sharding: {B * world} pairs over {world} GPUs)"),
                           "global_batch": B * world, "N": N, "M": M, "variant": args.variant,
                           "parallelism": f"batch-sharded x{world}", "gather": args.gather if multi else "none",
                           "backend": ("rccl" if backend == "nccl" else backend + (" (ranks share one GPU: test mode)" if shared else "")) if multi else "none",
                           "arith": "fwd: scaled exp-domain f32 (exact power-of-two rescaling); bwd: f32; adjoint pair: f64 carries; f32 storage",
                           "data_dependent": "the backward sweep does not run 32-step chunks whose outputs are exactly +0 nor read their state "
                                             "(bit-identical E; 47 % of E's cells, 29 % of the chunks on this data: DESIGN.md 3.8); the forward sweep "
                                             "-- the kernel the roofline is quoted on -- does all of its work"},
                "kernel_ms": ms,
                "kernel_ms_how": f"HIP event pairs on the launch stream around every 4th launch of each kernel, in {KSTEPS} steps of the same loop run right behind the timed region; the timed region itself holds no event record besides the pair around all of it",
                "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                             "real_frac": (traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                             "algorithmic_bytes_per_launch": dom_bytes,
                             "launch_ms": dom_ms,
                             "kernels": per_kernel,
                             # all sweeps of one step over the step's wall time: 12 + 12 B per cell for fwd + bwd, + 32 + 32 for the
                             # adjoint pair of the training modes
                             "whole_step_algorithmic_bytes": step_algo_bytes,
                             "whole_step_frac": (step_algo_bytes * args.steps / elapsed) / (HBM_PEAK_GBS * 1e9),
                             # ... and the same on the bytes the sweeps really moved (the counter passes' figures, when they belong to
                             # these sources): the figure that says how close the step is to the memory system
                             "whole_step_traffic": (sum(k_["traffic"] for k_ in per_kernel.values()) if per_kernel and all(k_["traffic"] for k_ in per_kernel.values()) else None),
                             "whole_step_real_frac": ((sum(k_["traffic"] for k_ in per_kernel.values()) * args.steps / elapsed) / (HBM_PEAK_GBS * 1e9)
                                                      if per_kernel and all(k_["traffic"] for k_ in per_kernel.values()) else None),
                             # the chip's own ceilings for bare streams of 1 KB per wave-instruction, one wave per SIMD (tools/ubench/
                             # vmemissue.hip, profiles/r06_ubench_vmemissue.txt): what "memory-bound" means on this part
                             "measured_stream_ceilings_GBs": {"loads": 7000.0, "stores": 5400.0, "fwd_mix": 5050.0}},
            }
            if pipelined is not None:
                line["pipelined_walk"] = pipelined
            if no_skip is not None:
                line["no_skip"] = no_skip
            if clocks is not None:
                line["clocks"] = clocks
            if dom.startswith("sdp_scores"):
                # the GEMM dominates this mode: its bound is the matrix pipe (filled in below as scores_roofline); the
                # HBM figures above then describe nothing and are replaced
                line["roofline"] = None
            if args.mode == "scores+dp" and ("sdp_scores_kernel" in ms or "sdp_scores_x6_kernel" in ms or "sdp_scores_x6w_kernel" in ms):
                flops = 2.0 * 2.0 * B * N * M * args.D          # two (N,D) x (D,M) products per pair
                if "sdp_scores_x6_kernel" in ms or "sdp_scores_x6w_kernel" in ms:
                    # three exact bf16 pieces per operand, six piece products per k: the pipe executes 6x the algorithmic
                    # flops; `achieved` / `peak` are what ran on the bf16 pipe, `algorithmic` the fp32-equivalent rate
                    x6name = "sdp_scores_x6w_kernel" if "sdp_scores_x6w_kernel" in ms else "sdp_scores_x6_kernel"
                    t_ms = ms[x6name]
                    tf = 6.0 * flops / (t_ms * 1e-3) / 1e12
                    line["scores_roofline"] = {"bound": "mfma", "kernel": x6name, "achieved": tf, "peak": 2516.6, "unit": "TFLOP/s", "traffic": None,
                                               "frac": tf / 2516.6, "dtype": "bf16 x 6 piece products (v_mfma_f32_32x32x16_bf16), f32 accumulate",
                                               "algorithmic": flops / (t_ms * 1e-3) / 1e12, "algorithmic_vs_f32_mfma_peak": flops / (t_ms * 1e-3) / 1e12 / 157.3,
                                               "D": args.D, "launch_ms": t_ms, "flops_per_launch": flops}
                else:
                    tf = flops / (ms["sdp_scores_kernel"] * 1e-3) / 1e12
                    line["scores_roofline"] = {"bound": "mfma", "kernel": "sdp_scores_kernel", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "traffic": None,
                                               "frac": tf / 157.3, "dtype": "f32 (v_mfma_f32_32x32x2_f32)", "D": args.D,
                                               "launch_ms": ms["sdp_scores_kernel"], "flops_per_launch": flops}
            if args.mode == "scores+dp":
                # the stated baseline for the scores: the reference's own two lines (alignment.py:122-123) as PyTorch runs
                # them on this box (rocBLAS/hipBLASLt batched GEMM + elementwise kernels), same inputs, outside the timed region
                import torch.nn.functional as F

                def ref_scores():
                    return (F.softplus(torch.einsum('bid,bjd->bij', emb[0], emb[1])),
                            F.logsigmoid(torch.einsum('bid,bjd->bij', emb[2], emb[3])))
                ref_scores()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    ref_scores()
                e1.record()
                torch.cuda.synchronize()
                line["scores_torch_baseline"] = {"ms": e0.elapsed_time(e1) / 5, "what": "F.softplus(torch.einsum('bid,bjd->bij', zx, zy)), "
                                                 "F.logsigmoid(torch.einsum(...gx, gy)) (deepblast/alignment.py:122-123), fp32, this box"}
            if line["roofline"] is None:
                line["roofline"] = line.get("scores_roofline")
            if args.mode == "fwdbwd":
                line["config"]["step"] = "Vt = dec(theta, A); Vt.sum().backward()  (public API, autograd, .grad accumulation included)"
            if direct is not None:
                # the same two sweeps asked for with torch.autograd.grad(Vt, theta, ones): no sum / fill kernels, no .grad
                line["direct_cotangent"] = {"ms_per_step": direct * 1e3, "value": world * per_step_updates / direct,
                                            "step": "Vt = dec(theta, A); torch.autograd.grad(Vt, theta, ones)"}
            if e_gather is not None:
                line["with_e_gather"] = {"ms_per_step": e_gather * 1e3, "value": world * per_step_updates / e_gather,
                                         "bytes_into_each_gpu": (world - 1) * B * N * M * 4,
                                         "overlap": "chunked" if args.e_chunks > 1 else "none", "e_chunks": args.e_chunks}
                if e_gather_one is not None:
                    line["with_e_gather"]["one_collective_ms_per_step"] = e_gather_one * 1e3
            if paths_gather is not None:
                line["with_paths_gather"] = {"ms_per_step": paths_gather * 1e3, "value": world * per_step_updates / paths_gather,
                                             "bytes_into_each_gpu": (world - 1) * B * (N + M + 4) * 4}
            if "sdp_traceback_kernel" in ms:
                line["traceback_ms"] = ms["sdp_traceback_kernel"]
            if world == 1 and not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(args)
                line["cpu_baseline"]["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
            print(json.dumps(line), flush=True)

    # N > 1: the same job with the expected-alignment matrices gathered as well (report scaling with
    # and without the E gather); a secondary figure, never `value`
    e_gather = e_gather_one = None
    # (a secondary figure must not cost the primary one: an error here -- the same on every rank, e.g. out of memory for
    # the gathered E -- is reported on stderr and the line goes out without that field)
    def secondary(kind, e_chunks=None, idiom=None, nsteps=None):
        nsteps = nsteps or min(args.steps, 20)    # (the gathers of E move gigabytes per step: a bounded region)
        try:
            aligner.gather = kind
            aligner.e_chunks = args.e_chunks if e_chunks is None else e_chunks
            aligner.idiom = idiom or aligner.idiom
            if os.environ.get("BENCH_TEST_HANG_RANK") == str(rank):   # test hook: this rank never reaches the collective
                time.sleep(3600)
            step()
            dt_s, _ = timed(nsteps)
            return dt_s / nsteps
        except Exception as ex:   # noqa: BLE001
            print(f"[bench] rank {rank}: secondary measurement gather={kind!r} failed: {ex}", file=sys.stderr, flush=True)
            return None
        finally:
            aligner.gather = args.gather
            aligner.e_chunks = args.e_chunks
            aligner.idiom = "sum_backward" if args.mode == "fwdbwd" else "grad"

    # A secondary figure must not cost the primary one either by HANGING: a rank that fails alone leaves the others
    # inside a collective.  A watchdog on every rank lets rank 0 print the line without the secondary fields and ends the
    # process if the secondary measurements have not finished in time (nothing here has ever run on more than one GPU).
    import threading
    emitted = threading.Lock()

    def watchdog():
        print(f"[bench] rank {rank}: secondary measurements did not finish in time: the line goes out without them", file=sys.stderr, flush=True)
        if emitted.acquire(blocking=False):
            emit(None, None, None)
        sys.stdout.flush()
        os._exit(0)

    dog = None
    if multi and args.mode == "fwdbwd" and not os.environ.get("BENCH_NO_SECONDARY"):
        # (BENCH_WATCHDOG_S: test hook, tests/test_multirank_one_gpu.py fires the watchdog on purpose)
        dog = threading.Timer(float(os.environ.get("BENCH_WATCHDOG_S") or max(60.0, 200.0 * elapsed)), watchdog)
        dog.daemon = True
        dog.start()
    direct = None
    if args.mode == "fwdbwd" and not os.environ.get("BENCH_NO_SECONDARY"):
        direct = secondary(args.gather if multi else "none", idiom="grad", nsteps=args.steps)   # (the primary's own step count: like for like)
    # the control of the data-dependent saving: the same step with the backward sweep running EVERY chunk
    # (variant | SDP_NO_ZERO_SKIP; E is bit-identical) -- a number that moves with the data travels with its control
    if rank == 0 and not multi and args.mode in ("fwdbwd", "train") and not os.environ.get("BENCH_NO_SECONDARY"):
        try:
            eng.zero_skip = False
            step()
            dt_ns, _ = timed(args.steps)   # (the primary's own step count, no instruments: a like-for-like control)
            nm = kernel_loop(KernelTimer(every=timer.every))
            no_skip = {"ms_per_step": dt_ns / args.steps * 1e3, "value": per_step_updates * args.steps / dt_ns,
                       "bwd_ms": next((v for k, v in nm.items() if k.startswith("sdp_bwd")), None),
                       "fwd_ms": next((v for k, v in nm.items() if k.startswith("sdp_fwd")), None), "steps": args.steps,
                       "what": "the same step with variant | SDP_NO_ZERO_SKIP: the backward sweep runs every chunk and reads all of its state (bit-identical E)"}
        except Exception as ex:   # noqa: BLE001
            print(f"[bench] no_skip control failed: {ex}", file=sys.stderr, flush=True)
        finally:
            eng.zero_skip = True
            eng.launch_hook = None
    # inference as a pipeline: a walk is one wavefront per pair, bound by memory latency, and needs 4 KB of LDS -- it fits beside the
    # next batch's sweeps on every CU.  Same step count, same fences; every walk is inside the timed region (the closing fence
    # synchronises the device, i.e. both streams)
    if rank == 0 and not multi and args.mode == "align+traceback" and not os.environ.get("BENCH_NO_SECONDARY"):
        try:
            walk["stream"] = torch.cuda.Stream()
            step()
            dt_pw, _ = timed(args.steps)
            pipelined = {"ms_per_step": dt_pw / args.steps * 1e3, "value": per_step_updates * args.steps / dt_pw, "steps": args.steps,
                         "what": "the same step with the batched walk of batch k on a second HIP stream, beside the forward sweep of batch k + 1"}
        except Exception as ex:   # noqa: BLE001
            print(f"[bench] pipelined_walk failed: {ex}", file=sys.stderr, flush=True)
        finally:
            walk["stream"] = None
    # (The same with the scores GEMM of batch k + 1 on a second stream beside the sweeps of batch k was measured in round 6 and is
    #  no gain: 0.984 -> 0.972 ms per step -- the GEMM's workgroups fill the chip and the power budget; not kept.)
    if not os.environ.get("BENCH_NO_CLOCKS") and not os.environ.get("BENCH_NO_SECONDARY"):
        # every rank loops the same number of steps (~1.5 s by the timed region's own figure, the max over ranks); rank 0 samples
        rounds = max(1, min(400, int(1.5 / max(50 * elapsed / args.steps, 1e-4))))
        busy = lambda: [step() for _ in range(50)]
        try:
            if rank == 0:
                clocks = ClockSampler(local_rank).run(1.5, busy, rounds)
            else:
                for _ in range(rounds):
                    busy()
                    torch.cuda.synchronize()
        except Exception as ex:   # noqa: BLE001
            print(f"[bench] clock sampling failed: {ex}", file=sys.stderr, flush=True)
    if multi and args.mode == "fwdbwd" and args.gather != "e" and not os.environ.get("BENCH_NO_SECONDARY"):
        e_gather = secondary("e")                 # backward sweep + gather in --e-chunks pieces
        e_gather_one = secondary("e", e_chunks=1) if args.e_chunks > 1 else None   # one collective after the sweep
    # ... and with the tracebacks gathered instead (device walk + (N+M+4) int32 per pair over the wire)
    paths_gather = None
    if multi and args.mode == "fwdbwd" and args.gather != "paths" and not os.environ.get("BENCH_NO_SECONDARY"):
        paths_gather = secondary("paths")

    if dog is not None:
        dog.cancel()
    if emitted.acquire(blocking=False):
        emit(e_gather, e_gather_one, paths_gather, direct)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
