"""Debug / timing of the parts schedule: forward Vt and E with parts against the same library with parts switched off (experiments build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import datagen, gpu_tune
B, N, M = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (5, 1024, 1024)))
use_lens = "lens" in sys.argv
exp = gpu_tune.load(os.path.join(ROOT, "deepblast_amd", "libsdp_hip_exp.so"))
cfg3 = "cfg3" in sys.argv   # BASELINE configs[2] exactly as tools/gpu_configs.py builds it
if cfg3:
    ln3 = datagen.lengths(2, B, 64, 1024)
    N, M = int(ln3[:, 0].max()), int(ln3[:, 1].max())
theta, A = datagen.theta_A(2 if cfg3 else 91000 + N, B, N, M)
t, a = torch.from_numpy(theta).cuda(), torch.from_numpy(A).cuda()
et = torch.ones(B, device="cuda")
lens = None
if use_lens:
    ln = datagen.lengths(2, B, 64, N); ln[:, 1] = np.minimum(ln[:, 1], M)
    if cfg3:
        ln = ln3
    lens = torch.from_numpy(ln).cuda()
stream = torch.cuda.current_stream().cuda_stream
res = {}
for mask in (64, 512 | 128, 512 | 256, 512):
    gpu_tune.set_debug(exp, mask)
    st = torch.empty(exp.sdp_state_bytes(B, N, M) // 4, device="cuda")
    vt = torch.empty(B, device="cuda"); E = torch.empty(B, N, M, device="cuda")
    lp = None if lens is None else lens.data_ptr()
    f = lambda: exp.sdp_forward_f32(t.data_ptr(), a.data_ptr(), st.data_ptr(), vt.data_ptr(), B, N, M, lp, 0, 0, stream)
    g = lambda: exp.sdp_backward_f32(et.data_ptr(), st.data_ptr(), E.data_ptr(), B, N, M, lp, 0, 0, stream)
    assert f() == 0 and g() == 0
    torch.cuda.synchronize()
    tf, tb = gpu_tune.timeit(f), gpu_tune.timeit(g)
    res[mask] = (vt.clone(), E.clone(), st.clone())
    print(f"parts {({64: 'off', 512 | 128: 'fwd only', 512 | 256: 'bwd only', 512: 'on'})[mask]}: fwd {tf:.1f} us  bwd {tb:.1f} us")
gpu_tune.set_debug(exp, 0)
for mask in (512 | 128, 512 | 256, 512):
    dv = (res[mask][0] - res[64][0]).abs()
    dE = (res[mask][1] - res[64][1]).abs()
    rows = dE.amax(dim=2)[0].cpu().numpy()
    cols = dE.amax(dim=1)[0].cpu().numpy()
    bad, badc = np.nonzero(rows > 0)[0], np.nonzero(cols > 0)[0]
    print(f"mask {mask}: Vt diff {dv.cpu().numpy()[:4]}; pair 0 rows of E that differ:", (bad.min(), bad.max(), len(bad)) if len(bad) else "none",
          "cols", (badc.min(), badc.max(), len(badc)) if len(badc) else "none", " max", float(dE.max()))

# packed state: per (pair, strip) tpad * 384 bytes; compare parts-forward (mask 128) with no parts (64), pair 0
nstrips, tpad = (N + 63) // 64, (M + 63 + 63) // 64 * 64
sa = res[512 | 128][2].view(torch.uint8)[: nstrips * tpad * 384].view(nstrips, tpad // 2, 768).cpu().numpy()
sb = res[64][2].view(torch.uint8)[: nstrips * tpad * 384].view(nstrips, tpad // 2, 768).cpu().numpy()
for s_ in range(nstrips):
    d = (sa[s_] != sb[s_]).any(axis=1)
    idx = np.nonzero(d)[0]
    print(f"strip {s_}: record rows (2 steps each) that differ: {len(idx)} of {tpad // 2}; first {idx[:6]}")
    if len(idx):
        r = idx[0]
        lanes = np.nonzero((sa[s_, r].reshape(64, 12) != sb[s_, r].reshape(64, 12)).any(axis=1))[0]
        print("    lanes differing in the first such row:", lanes[:10], "...", len(lanes))
