#!/bin/bash
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_robustness_gpu.py -m gpu -x -q 2>&1 | tail -6
