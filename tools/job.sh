#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
bash tools/gpu_sq.sh sqr2 256 > gpurun_out/sq_r2.txt 2>&1; tail -12 gpurun_out/sq_r2.txt
