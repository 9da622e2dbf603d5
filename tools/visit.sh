mkdir -p gpurun_out/adj2
export TMPDIR=/tmp
(timeout 300 python tools/adj_trace.py b 2>&1 | grep -v amdgpu | head -11) | tee gpurun_out/adj2/abwd_trace.txt
(ALIAS=7 timeout 300 python tools/adj_trace.py b 2>&1 | grep -v amdgpu | head -11) | tee gpurun_out/adj2/abwd_trace_alias7.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/adj2/tests.txt
