#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_scores.py tests/test_end_to_end_gpu.py tests/test_abi.py -m gpu -x -q < /dev/null 2>&1 | tail -6
timeout 300 python bench.py --mode scores+dp --steps 20 --warmup 3 --no-cpu-baseline < /dev/null 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print(d['ms_per_step'], d['scores_roofline'])"
