// Shader clock under a sustained fp32 MFMA load (what the scores kernel runs): every wave issues v_mfma_f32_32x32x2_f32
// back to back and reads s_memtime (shader clock) and s_memrealtime (100 MHz) around the loop.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_clock.hip -o tools/ubench/mfma_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void __launch_bounds__(256) burn(float *out, long long *clk, int iters)
{
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a)
        for (int v = 0; v < 16; ++v) acc[a][v] = 0.f;
    const float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    const long long c0 = clock64(), r0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    const long long c1 = clock64(), r1 = wall_clock64();
    float s = 0.f;
    for (int a = 0; a < 4; ++a)
        for (int v = 0; v < 16; ++v) s += acc[a][v];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[2 * blockIdx.x] = c1 - c0, clk[2 * blockIdx.x + 1] = r1 - r0;
}
int main()
{
    const int wgs = 256 * 4;   // 4 workgroups of 4 waves per CU: 4 waves per SIMD, like the scores kernel
    float *out; long long *clk;
    CHECK(hipMalloc(&out, wgs * 256 * 4)); CHECK(hipMalloc(&clk, wgs * 16));
    long long *h = (long long *)malloc(wgs * 16);
    for (int rep = 0; rep < 6; ++rep) {
        const int iters = 4000;   // 16 MFMAs per iteration and wave
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(burn, dim3(wgs), dim3(256), 0, 0, out, clk, iters); CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h, clk, wgs * 16, hipMemcpyDeviceToHost));
        double sc = 0, rt = 0; for (int i = 0; i < wgs; ++i) sc += h[2 * i], rt += h[2 * i + 1];
        const double flop = (double)wgs * 4 * iters * 16 * 4096.0;
        printf("rep %d: %.3f ms, %.1f TFLOP/s, shader clock %.0f MHz (s_memtime ticks per 100 MHz tick x 100), cycles per MFMA per SIMD %.1f\n", rep, ms,
               flop / ms / 1e9, sc / rt * 100.0, (sc / wgs) / (iters * 16.0 * 4));
    }
    return 0;
}
