"""Single-rank RCCL sanity check of ShardedAligner's asynchronous all-gather (run under torch.distributed.run)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, torch.distributed as dist, datagen
from deepblast_amd import NeedlemanWunschDecoder
from deepblast_amd.distributed import ShardedAligner
rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
B, N, M = 256, 512, 512
th, A = datagen.theta_A(1, B, N, M)
theta = torch.from_numpy(th).cuda(); a = torch.from_numpy(A).cuda()
for gather in ("none", "vt"):
    al = ShardedAligner(NeedlemanWunschDecoder("softmax"), gather=gather)
    for _ in range(3): out = al.align(theta, a)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): out = al.align(theta, a)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20 * 1e3
    ok = out["Vt"] is None or torch.equal(out["Vt"], out["Vt_local"])
    print(f"gather={gather}: {dt:.4f} ms/step  gathered==local: {ok}")
dist.destroy_process_group()
