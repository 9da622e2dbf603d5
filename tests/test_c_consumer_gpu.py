"""GPU: the C ABI used the way a non-Python host would use it -- tests/c_abi/consumer.cpp: hipMalloc'd buffers, a stream of
its own, include/sdp.h, no PyTorch in the process -- built here with the host compiler (g++) against the in-tree library and run; it checks
the four sweeps against the oracle (linked into the TEST binary only) and the walks for consistency."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("variant", ["nw", "sw"])
def test_c_consumer_of_the_abi(tmp_path, variant):
    cxx = shutil.which("g++")
    if cxx is None or not os.path.isdir("/opt/rocm/include"):
        pytest.skip("no host compiler / ROCm headers on this box")
    from deepblast_amd import build
    from oracle import oracle
    lib = build.build()
    oracle.build()
    exe = str(tmp_path / "consumer")
    libdir, odir = os.path.dirname(lib), os.path.join(ROOT, "oracle")
    # the HOST compiler: nothing in the consumer is device code
    cmd = [cxx, "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "c_abi", "consumer.cpp"),
           "-I" + os.path.join(ROOT, "include"), "-L" + libdir, "-l:" + os.path.basename(lib), "-L" + odir, "-l:liboracle.so",
           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + libdir, "-Wl,-rpath," + odir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, variant], capture_output=True, text=True, timeout=300)
    print(r.stdout)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-2000:] + r.stderr[-2000:]
