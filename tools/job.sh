#!/bin/bash
OUT=gpurun_out/r03a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_distributed_nccl_gpu.py tests/test_comm_gpu.py -m gpu -x -q 2>&1 | tail -8 > $OUT/pytest3.txt; cat $OUT/pytest3.txt
