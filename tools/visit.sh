mkdir -p gpurun_out/v1
export TMPDIR=/tmp
(timeout 600 python tools/bitcmp.py ref=r05 sw x 2>&1 | grep -v amdgpu | tail -30) > gpurun_out/v1/bitcmp.txt
(timeout 600 python tools/bitcmp.py ref=r05 sw lens 2>&1 | grep -v amdgpu | tail -20) >> gpurun_out/v1/bitcmp.txt
(timeout 600 python tools/steady.py 256x512x512 64x512x512 256x1024x1024 2>&1 | grep -v amdgpu | tail -20) > gpurun_out/v1/steady.txt
(timeout 600 python tools/steady.py 256x512x512 ADJ=1 2>&1 | grep -v amdgpu | tail -8) >> gpurun_out/v1/steady.txt
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/v1/pytest.txt
cat gpurun_out/v1/bitcmp.txt gpurun_out/v1/steady.txt gpurun_out/v1/pytest.txt
