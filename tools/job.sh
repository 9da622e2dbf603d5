#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/pytest.txt
tail -3 gpurun_out/pytest.txt
timeout 600 python tools/ab.py 256x512x512 64x512x512 256x1024x1024 > gpurun_out/ab.txt 2>&1; cat gpurun_out/ab.txt
