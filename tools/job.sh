#!/bin/bash
OUT=gpurun_out/r03a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $OUT/pytest2.txt; cat $OUT/pytest2.txt
