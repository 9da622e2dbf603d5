mkdir -p gpurun_out/final
export TMPDIR=/tmp
(timeout 1200 python tools/tb_fuzz.py 120 2026 2>&1 | grep -v amdgpu | tail -5) | tee gpurun_out/final/tb_fuzz.txt
