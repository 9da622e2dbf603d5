#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/ab.py 256x512x512 adj < /dev/null 2>&1 | grep -v amdgpu.ids | tail -4
SDP_LIB_PATH=$PWD/build_variants/libsdp_abwd32.so FUZZ_REPORT=4e-5 timeout 900 python tools/fuzz2.py 150 4242 < /dev/null 2>&1 | grep -v amdgpu.ids | tail -6
SDP_LIB_PATH=$PWD/build_variants/libsdp_abwd32.so timeout 600 python tools/steep_probe.py < /dev/null 2>&1 | grep -A2 "^(" | grep -v "^--" | head -20
