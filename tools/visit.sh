mkdir -p gpurun_out/genf
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/genf/tests2.txt
timeout 600 python tools/steady.py 256x1022x1020 64x1022x1020 256x500x516 256x512x512 2>&1 | grep "B=" | tee gpurun_out/genf/steady_gen2.txt
timeout 300 python tools/gpu_configs.py 2> /dev/null | tee gpurun_out/genf/configs2.txt
