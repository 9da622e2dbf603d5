mkdir -p gpurun_out/tb
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_autograd_gpu.py tests/test_call_sites_gpu.py tests/test_multirank_one_gpu.py -q -m gpu -k "traceback or paths or call_site" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -4 | tee gpurun_out/tb/tests.txt
timeout 300 python tools/tb_probe.py 2>&1 | grep "B=" | tee gpurun_out/tb/tb_probe.txt
