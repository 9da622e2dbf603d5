#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 120 tools/ubench/mfma_clock < /dev/null 2>&1 | tail -8
