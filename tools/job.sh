#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "fuzz" 2>&1 | grep -E "passed|failed|Error|FAILED" | tail -3
timeout 1200 python tools/fuzz2.py 700 999 > gpurun_out/fuzz2b.txt 2>&1; tail -4 gpurun_out/fuzz2b.txt
