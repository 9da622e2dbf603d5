mkdir -p gpurun_out/v12
export TMPDIR=/tmp
(timeout 900 python tools/steady.py 256x512x512 64x512x512 2>&1 | grep -v amdgpu | tail -8) | tee gpurun_out/v12/steady.txt
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee gpurun_out/v12/pytest.txt
