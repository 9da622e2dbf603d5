#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_distributed_nccl_gpu.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_dist.txt
cat gpurun_out/pytest_dist.txt
