#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/lens_probe.py > gpurun_out/lens_probe.txt 2>&1; cat gpurun_out/lens_probe.txt
