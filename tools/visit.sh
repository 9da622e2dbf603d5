mkdir -p gpurun_out/soak3
export TMPDIR=/tmp
(timeout 1500 python tools/fuzz2.py 2500 621 2>&1 | grep -v amdgpu | tail -12) | tee gpurun_out/soak3/fuzz2_2500.txt
(timeout 900 python tools/fuzz2.py 120 622 full 2>&1 | grep -v amdgpu | tail -6) | tee gpurun_out/soak3/fuzz2_full120.txt
(timeout 600 python tools/parts_fuzz.py 400 2>&1 | grep -v amdgpu | tail -3) | tee gpurun_out/soak3/parts_fuzz_400.txt
