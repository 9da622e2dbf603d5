mkdir -p gpurun_out/v10
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -15 | tee gpurun_out/v10/pytest.txt
