#!/bin/bash
OUT=gpurun_out/r03a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_scores.py tests/test_abi.py -m gpu -x -q 2>&1 | tail -3
timeout 600 python tools/scores_probe2.py 2>&1 | grep -v amdgpu
timeout 300 python bench.py --mode scores+dp --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['kernel_ms'], d['scores_torch_baseline']['ms'], d['roofline']['frac'])"
