#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/scores_probe.py < /dev/null 2>&1 | grep -v amdgpu.ids | tail -9
