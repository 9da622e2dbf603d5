#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/ab.py 256x512x512 adj 2>&1 | tail -2
timeout 600 python tools/ab.py 256x512x512 adj 2>&1 | tail -2
