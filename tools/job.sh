#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for m in train train-mce train-mce-fused; do
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --mode $m 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$m', round(d['ms_per_step'],4), {k: round(v,4) for k,v in d['kernel_ms'].items()})"
done
