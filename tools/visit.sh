mkdir -p gpurun_out/alias
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 tools/ubench/vmemissue.hip -o /tmp/vmemissue && (timeout 300 /tmp/vmemissue 2>&1) | tee gpurun_out/alias/vmemissue5.txt
