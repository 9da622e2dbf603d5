mkdir -p gpurun_out/v4
export TMPDIR=/tmp
hipcc --offload-arch=gfx950 -O3 -o /tmp/vmemissue tools/ubench/vmemissue.hip 2> /dev/null && timeout 300 /tmp/vmemissue 2>&1 | tee gpurun_out/v4/vmemissue.txt
