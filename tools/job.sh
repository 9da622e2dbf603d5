#!/bin/bash
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python tools/gpu_configs.py 2>&1 | tail -4
python bench.py 2>&1 | tail -1 | cut -c1-400
