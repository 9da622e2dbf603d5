#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error" | tail -3
