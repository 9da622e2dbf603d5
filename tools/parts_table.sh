#!/bin/bash
# The parts policy, measured: one workgroup per pair against pairs spread over several workgroups, forward / backward, over shapes (-> profiles/r05_parts_table.txt)
mkdir -p gpurun_out
for args in "256 1022 1020 lens cfg3" "256 640 640 lens" "256 512 512 lens" "128 512 512 lens" "64 1022 1020 lens" "16 1024 1024" "64 1024 512" "64 640 500" "32 2000 1000 lens"; do
echo "== $args"; timeout 200 python tools/parts_probe.py $args 2>/dev/null | grep "^parts"
done
