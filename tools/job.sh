#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/waves_probe.py < /dev/null 2>&1 | grep -v amdgpu.ids | tail -16
