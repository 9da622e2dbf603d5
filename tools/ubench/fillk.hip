// A persistent zero-fill kernel with a chosen number of workgroups (no LDS), for tools/fill_overlap_probe.py:
// does a slow fill next to the backward sweep overlap with it?   build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o /tmp/libfillk.so tools/ubench/fillk.hip
#include <hip/hip_runtime.h>
__global__ void __launch_bounds__(256) fillk(float4 *p, size_t n16)
{
    const float4 z = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = z;
}
extern "C" int fill_launch(void *p, size_t n16, int blocks, void *stream)
{
    hipLaunchKernelGGL(fillk, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (float4 *)p, n16);
    return (int)hipGetLastError();
}
