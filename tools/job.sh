#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8
