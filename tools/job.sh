#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "long" < /dev/null 2>&1 | tail -6
