mkdir -p gpurun_out/v14
export TMPDIR=/tmp
timeout 600 python tools/lens_ab.py 2>&1 | grep -v amdgpu | tee gpurun_out/v14/lens_ab.txt
timeout 900 python -m pytest tests/test_robustness_gpu.py::test_what_lies_beside_the_matrix_takes_no_part_in_anything tests/test_fuzz_gpu.py -q -s 2>&1 | grep -v "amdgpu" | grep "^E   .*Error\|passed\|failed\|FAILED\|fuzz3\|thin pairs" | head -30 | tee gpurun_out/v14/tests.txt
(timeout 600 python tools/steady.py 256x512x512 64x512x512 2>&1 | grep -v amdgpu | tail -6) | tee gpurun_out/v14/steady.txt
