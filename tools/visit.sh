mkdir -p gpurun_out/genf
export TMPDIR=/tmp
timeout 600 python tools/steady.py 256x1022x1020 64x1022x1020 256x500x516 64x500x516 2>&1 | grep "B=" | tee gpurun_out/genf/steady_gen3.txt
timeout 300 python tools/lens_ab.py 2>&1 | grep -v amdgpu | tail -3 | tee gpurun_out/genf/lens_ab4.txt
timeout 300 python tools/lens_ab.py nofill 2>&1 | grep -v amdgpu | tail -3 | tee gpurun_out/genf/lens_ab4_nofill.txt
timeout 300 python tools/gpu_configs.py 2> /dev/null | tee gpurun_out/genf/configs3.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -5 | tee gpurun_out/genf/tests3.txt
