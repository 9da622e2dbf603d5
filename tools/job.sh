#!/bin/bash
OUT=gpurun_out/round_r02x; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sc -o sc -- python bench.py --mode scores+dp --steps 10 --warmup 2 --no-cpu-baseline < /dev/null > $OUT/bench_sc.json 2> $OUT/sc.err
head -4 $OUT/prof_sc/sc_kernel_stats.csv | cut -c1-140; tail -1 $OUT/bench_sc.json | cut -c1-160
