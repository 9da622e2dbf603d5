"""Would the zero fill of E overlap with the backward sweep if it came from a kernel of its own (no LDS: its workgroups fit on
CUs next to the sweep's) on a second stream?  Emulation: the backward sweep of BASELINE configs[2] from a build that does not
fill (build_variants/libsdp_nz.so, wrong E outside the blocks) + a torch fill of as many bytes on a side stream."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import datagen, gpu_tune
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ln3 = datagen.lengths(2, B, 64, 1024)
N, M = int(ln3[:, 0].max()), int(ln3[:, 1].max())
theta, A = datagen.theta_A(2, B, N, M)
t, a = torch.from_numpy(theta).cuda(), torch.from_numpy(A).cuda()
et = torch.ones(B, device="cuda")
lens = torch.from_numpy(ln3).cuda()
fill_bytes = int((B * N * M - (ln3[:, 0].astype(np.int64) * ln3[:, 1]).sum()) * 4)
dummy = torch.empty(fill_bytes // 4, device="cuda")
side = torch.cuda.Stream()
main = gpu_tune.load(os.path.join(ROOT, "deepblast_amd", "libsdp_hip.so"))
nz = gpu_tune.load(os.path.join(ROOT, "build_variants", "libsdp_nz.so"))
stream = torch.cuda.current_stream().cuda_stream
def setup(lib):
    st = torch.empty(lib.sdp_state_bytes(B, N, M) // 4, device="cuda")
    vt = torch.empty(B, device="cuda"); E = torch.empty(B, N, M, device="cuda")
    assert lib.sdp_forward_f32(t.data_ptr(), a.data_ptr(), st.data_ptr(), vt.data_ptr(), B, N, M, lens.data_ptr(), 0, 0, stream) == 0
    return lambda: lib.sdp_backward_f32(et.data_ptr(), st.data_ptr(), E.data_ptr(), B, N, M, lens.data_ptr(), 0, 0, stream)
b_main, b_nz = setup(main), setup(nz)
def both():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        dummy.zero_()
    b_nz()
    torch.cuda.current_stream().wait_stream(side)
print(f"B={B} {N}x{M}: fill {fill_bytes / 1e6:.0f} MB")
print(f"backward with its own fill (shipped)        : {gpu_tune.timeit(b_main):7.1f} us")
print(f"backward without fill                        : {gpu_tune.timeit(b_nz):7.1f} us")
print(f"torch fill of the same bytes alone           : {gpu_tune.timeit(lambda: dummy.zero_()):7.1f} us")
print(f"backward without fill || torch fill (2 streams): {gpu_tune.timeit(both):7.1f} us")

# ... and with a fill kernel of a few persistent workgroups (tools/ubench/fillk.hip), which cannot crowd the sweep off the CUs
import ctypes, subprocess
so = "/tmp/libfillk.so"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "ubench", "fillk.hip")])
fk = ctypes.CDLL(so)
fk.fill_launch.restype, fk.fill_launch.argtypes = ctypes.c_int, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
for blocks in (16, 32, 64, 128, 256, 1024):
    def fill_alone():
        assert fk.fill_launch(dummy.data_ptr(), fill_bytes // 16, blocks, stream) == 0
    def both2():
        side.wait_stream(torch.cuda.current_stream())
        assert fk.fill_launch(dummy.data_ptr(), fill_bytes // 16, blocks, side.cuda_stream) == 0
        b_nz()
        torch.cuda.current_stream().wait_stream(side)
    print(f"fill kernel of {blocks:4d} workgroups: alone {gpu_tune.timeit(fill_alone):7.1f} us   next to the backward sweep {gpu_tune.timeit(both2):7.1f} us", flush=True)
