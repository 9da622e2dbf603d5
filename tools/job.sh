#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/ab.py 256x512x512 adj > gpurun_out/ab.txt 2>&1; cat gpurun_out/ab.txt
