#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/scores_probe.py < /dev/null 2>&1 | grep -v amdgpu.ids | tail -4
timeout 600 python -m pytest tests/test_scores.py -m gpu -x -q < /dev/null 2>&1 | tail -2
