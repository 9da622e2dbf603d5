#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/pytest.txt
tail -4 gpurun_out/pytest.txt
timeout 300 python bench.py --steps 20 --warmup 3 --mode train --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train', d['ms_per_step'], d['kernel_ms'])"
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fwdbwd', d['ms_per_step'], d['kernel_ms'])"
