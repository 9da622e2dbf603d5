#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8
for i in 1 2 3 4; do python bench.py --steps 100 --warmup 10 2>/dev/null | tail -1 | cut -c1-150; done
