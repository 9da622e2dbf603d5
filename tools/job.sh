#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_distributed_nccl_gpu.py tests/test_autograd_gpu.py -m gpu -x -q < /dev/null 2>&1 | tail -4
timeout 300 python bench.py --mode align+traceback --steps 50 --warmup 5 --no-cpu-baseline < /dev/null 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print(d['metric'], d['value'], d['ms_per_step'], d.get('traceback_ms'))"
