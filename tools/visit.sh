export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_robustness_gpu.py -q -m gpu -k "second_stream" 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -4
