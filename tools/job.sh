#!/bin/bash
OUT=gpurun_out/round_r02x; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_tb -o tb -- python bench.py --mode align+traceback --steps 10 --warmup 2 --no-cpu-baseline < /dev/null > $OUT/bench_tb.json 2> $OUT/tb.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sc -o sc -- python bench.py --mode scores+dp --steps 10 --warmup 2 --no-cpu-baseline < /dev/null > $OUT/bench_sc.json 2> $OUT/sc.err
for f in $(find $OUT -name "*kernel_stats.csv"); do echo "== $f"; head -5 $f | cut -c1-160; cp $f $OUT/$(basename $f); done
tail -1 $OUT/bench_tb.json | cut -c1-200; tail -1 $OUT/bench_sc.json | cut -c1-200
