#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 300 python tools/graph_probe.py 2>&1 | tail -3
timeout 300 python tools/graph_probe.py 2>&1 | tail -3
