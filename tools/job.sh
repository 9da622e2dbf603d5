#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2; do
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('hook  ', d['ms_per_step'], d['ms_per_step_median'], d['kernel_ms'])"
BENCH_NO_KERNEL_EVENTS=1 timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('nohook', d['ms_per_step'], d['ms_per_step_median'])"
done
