mkdir -p gpurun_out/tb
export TMPDIR=/tmp
timeout 600 python bench.py --mode scores+dp --no-cpu-baseline 2> gpurun_out/tb/bench_sc.err | tee gpurun_out/tb/bench_scores.json
tail -3 gpurun_out/tb/bench_sc.err
