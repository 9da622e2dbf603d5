#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/lens_probe.py 2>&1 | grep -E "padded shape|batch order|sorted"
