#!/bin/bash
OUT=gpurun_out/r03a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python tools/ab.py 256x512x512 2>&1 | grep -v amdgpu
timeout 600 python tools/ab.py 256x512x512 2>&1 | grep -v amdgpu
