#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
( echo "== tools/fuzz2.py 1000 2026 (mixed)"; FUZZ_REPORT=6e-5 timeout 2400 python tools/fuzz2.py 1000 2026 < /dev/null 2>&1 | grep -v amdgpu.ids | tail -12
  echo "== tools/fuzz2.py 250 5 full (full batches of mid-size odd shapes)"; FUZZ_REPORT=6e-5 timeout 1500 python tools/fuzz2.py 250 5 full < /dev/null 2>&1 | grep -v amdgpu.ids | tail -8 ) | tee gpurun_out/fuzz_extended.txt | tail -24
