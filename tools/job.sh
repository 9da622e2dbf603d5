#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_scores.py -m gpu -x -q 2>&1 | tail -12 > gpurun_out/pytest_scores.txt
tail -6 gpurun_out/pytest_scores.txt
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --mode scores+dp 2>&1 | tail -2 > gpurun_out/bench_scores.json
python -c "
import json; d=json.loads(open('gpurun_out/bench_scores.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['kernel_ms'], d.get('scores_roofline'))"
