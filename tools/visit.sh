mkdir -p gpurun_out/adj2
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 tools/ubench/f2mix.hip -o /tmp/f2mix && (timeout 300 /tmp/f2mix 2>&1) | tee gpurun_out/adj2/f2mix.txt
