#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/gpu_configs.py 2>&1 | grep configs
timeout 300 python tools/lens_probe.py 2>&1 | grep -E "padded shape|batch order, auto"
