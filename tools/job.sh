#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|FAILED" | tail -3
timeout 900 python tools/ab.py 256x512x512 256x512x500 2>&1 | grep "B="
timeout 900 python tools/fuzz2.py 300 > gpurun_out/fuzz2.txt 2>&1; tail -2 gpurun_out/fuzz2.txt
