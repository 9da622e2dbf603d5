#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 tools/ubench/isa_cost > gpurun_out/isa_cost2.txt 2>&1
timeout 900 python tools/abl_probe.py > gpurun_out/abl.txt 2>&1
cat gpurun_out/abl.txt
