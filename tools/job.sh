#!/bin/bash
export TMPDIR=/tmp
timeout 300 python tools/parts_probe2.py 2>&1 | grep -v amdgpu | tail -2
timeout 300 python tools/ab.py 256x512x512 256x1024x1024 16x1024x1024 | tail -3
for c in "256 1022 1020" "700 1022 1020" "256 640 640"; do echo "== lens $c"; timeout 300 python tools/parts_probe.py $c lens 2>&1 | grep "^parts"; done
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_robustness_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|FAILED"
