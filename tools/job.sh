#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|FAILED|assert" | tail -5
timeout 900 python tools/fuzz2.py 600 > gpurun_out/fuzz2.txt 2>&1; tail -5 gpurun_out/fuzz2.txt
