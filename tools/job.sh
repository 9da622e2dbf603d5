#!/bin/bash
mkdir -p gpurun_out/round_r02; export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q < /dev/null 2>&1 | grep -v amdgpu.ids | tail -15 | tee gpurun_out/round_r02/pytest_gpu.txt | grep -E "passed|failed"
