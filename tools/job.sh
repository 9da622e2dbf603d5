#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_comm_gpu.py -m gpu -x -q 2>&1 | tail -8
