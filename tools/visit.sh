mkdir -p gpurun_out/alias
export TMPDIR=/tmp
for d in 0 7 4 3; do echo "== DBG=$d (1 inputs, 2 outputs, 4 state of every pair aliased to pair 0's)"; timeout 300 python tools/steady.py 256x512x512 128x512x512 EXP=1 DBG=$d only=none ROUNDS=3 2>&1 | grep "B="; done | tee gpurun_out/alias/steady_alias.txt
