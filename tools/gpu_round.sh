#!/bin/bash
# One GPU-box visit: GPU test-suite, smoke, bench line, rocprofv3 kernel stats.
# Usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag> [quick]
TAG=${1:-r01}
mkdir -p gpurun_out
export TMPDIR=/tmp
if [ "$2" != "quick" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_gpu_$TAG.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke_$TAG.txt
fi
timeout 600 python bench.py --steps 20 --warmup 3 2> gpurun_out/bench_$TAG.err | tee gpurun_out/bench_$TAG.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_prof_$TAG.json 2> gpurun_out/rocprof_$TAG.err
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d gpurun_out/pmc_${C}_$TAG -o $TAG -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/pmc_${C}_$TAG.err
done
ls -R gpurun_out/prof_$TAG gpurun_out/pmc_FETCH_SIZE_$TAG | head -30
for f in $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv"); do echo "== $f"; cat $f; done
