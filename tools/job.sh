#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/sbprof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sbprof -o sb -- python tools/scores_bwd_probe.py > gpurun_out/sbprof/out.txt 2>&1
grep -v amdgpu gpurun_out/sbprof/out.txt | grep "us" 
f=$(find gpurun_out/sbprof -name "*kernel_stats.csv" | head -1); head -12 $f | cut -d, -f1-4 | cut -c1-120
find gpurun_out/sbprof -name "*kernel_trace.csv" -delete
