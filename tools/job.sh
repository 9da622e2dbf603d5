#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_distributed_nccl_gpu.py -m gpu -x -q < /dev/null 2>&1 | tail -3
