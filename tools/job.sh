#!/bin/bash
mkdir -p gpurun_out/round_r02; export TMPDIR=/tmp
timeout 1200 python tools/fuzz2.py 300 < /dev/null 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/round_r02/fuzz2.txt
