#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
for i in 1 2 3; do timeout 300 python bench.py --mode train --steps 40 --warmup 5 --no-cpu-baseline < /dev/null 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('%.4g' % d['value'], '%.4f ms' % d['ms_per_step'], {k: round(v, 3) for k, v in d['kernel_ms'].items()})"; done
