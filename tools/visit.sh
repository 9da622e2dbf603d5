mkdir -p gpurun_out/adj
export TMPDIR=/tmp
for a in 0 7 4 3; do
(ALIAS=$a timeout 300 python tools/adj_trace.py b 2>&1 | grep -v amdgpu | head -11) | tee gpurun_out/adj/abwd_trace_alias$a.txt
(ALIAS=$a timeout 300 python tools/adj_trace.py f 2>&1 | grep -v amdgpu | head -11) | tee gpurun_out/adj/afwd_trace_alias$a.txt
done
