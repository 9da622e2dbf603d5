// Memory-system microbenchmarks for the access patterns of the soft-DP kernels (gfx950).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mem.hip -o tools/ubench/mem
//  mode 0: every wave reads one 512-B row and writes one 512-B row per iteration (reads prefetched D deep)
//  mode 1: even waves only read (2 rows/iter), odd waves only write (2 rows/iter)  -- same total traffic
//  mode 2: read only     mode 3: write only
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int D, int mode>
__global__ void __launch_bounds__(256) stream(const float2 *in, float2 *out, int rows_per_wave)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const float2 *src = in + (size_t)wave * rows_per_wave * 64 + lane;
    float2 *dst = out + (size_t)wave * rows_per_wave * 64 + lane;
    // mode 1: readers and writers are separate WORKGROUPS' worth of waves decided by a uniform value
    const bool odd = __builtin_amdgcn_readfirstlane(wave) & 1;
    const bool do_r = mode == 0 || mode == 2 || (mode == 1 && !odd);
    const bool do_w = mode == 0 || mode == 3 || (mode == 1 && odd);
    float2 ring[D];
    float2 acc = make_float2(0.f, 0.f);
    if (do_r) {
#pragma unroll
        for (int d = 0; d < D; ++d) ring[d] = src[(size_t)d * 64];
    }
    for (int r0 = 0; r0 < rows_per_wave; r0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int r = r0 + d;
            float2 v = make_float2((float)r, (float)lane);
            if (do_r) {
                v = ring[d];
                const int rn = r + D < rows_per_wave ? r + D : r;
                ring[d] = src[(size_t)rn * 64];
                acc.x += v.x;
                acc.y += v.y;
            }
            if (do_w) dst[(size_t)r * 64] = v;
        }
    }
    if (acc.x == 12345.f) dst[0] = acc;
}

// 8 waves per workgroup: waves 0-3 stream reads, waves 4-7 stream writes (decoupled access by wave)
template <int D>
__global__ void __launch_bounds__(512) split8(const float2 *in, float2 *out, int rows_per_wave)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = blockIdx.x * 4 + (w & 3);
    const float2 *src = in + (size_t)wave * rows_per_wave * 64 + lane;
    float2 *dst = out + (size_t)wave * rows_per_wave * 64 + lane;
    if (w < 4) {
        float2 ring[D];
        float2 acc = make_float2(0.f, 0.f);
#pragma unroll
        for (int d = 0; d < D; ++d) ring[d] = src[(size_t)d * 64];
        for (int r0 = 0; r0 < rows_per_wave; r0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int r = r0 + d;
                const float2 v = ring[d];
                const int rn = r + D < rows_per_wave ? r + D : r;
                ring[d] = src[(size_t)rn * 64];
                acc.x += v.x;
                acc.y += v.y;
            }
        }
        if (acc.x == 12345.f) dst[0] = acc;
    } else {
        for (int r = 0; r < rows_per_wave; ++r) dst[(size_t)r * 64] = make_float2((float)r, (float)lane);
    }
}

int main(int argc, char **argv)
{
    const int nblocks = 256, rows = 1152;  // 1024 waves x 1152 rows x 512 B = 604 MB each way
    const size_t n = (size_t)nblocks * 4 * rows * 64;
    float2 *in, *out;
    CHECK(hipMalloc(&in, n * sizeof(float2)));
    CHECK(hipMalloc(&out, n * sizeof(float2)));
    CHECK(hipMemset(in, 0, n * sizeof(float2)));
    CHECK(hipMemset(out, 0, n * sizeof(float2)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const char *names[] = {"rw interleaved in every wave", "readers / writers split by wave", "read only", "write only"};
    auto bench = [&](auto kern, int mode, int D) {
        float best = 1e9;
        for (int pass = 0; pass < 3; ++pass) {
            CHECK(hipEventRecord(e0));
            for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(kern, dim3(nblocks), dim3(256), 0, 0, in, out, rows);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms / 5 < best ? ms / 5 : best;
        }
        const double bytes = (double)n * 8 * ((mode == 0) ? 2 : 1);
        printf("mode %d (%-32s) D=%2d: %7.1f us  %6.2f TB/s\n", mode, names[mode], D, best * 1e3, bytes / (best * 1e-3) / 1e12);
    };
    bench(stream<8, 0>, 0, 8);
    bench(stream<32, 0>, 0, 32);
    bench(stream<8, 1>, 1, 8);
    bench(stream<32, 1>, 1, 32);
    bench(stream<8, 2>, 2, 8);
    bench(stream<32, 2>, 2, 32);
    bench(stream<64, 2>, 2, 64);
    bench(stream<32, 3>, 3, 32);
    {
        float best = 1e9;
        for (int pass = 0; pass < 3; ++pass) {
            CHECK(hipEventRecord(e0));
            for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(split8<8>, dim3(nblocks), dim3(512), 0, 0, in, out, rows);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms / 5 < best ? ms / 5 : best;
        }
        printf("split8 (4 reader + 4 writer waves per CU, same bytes as mode 0) D= 8: %7.1f us  %6.2f TB/s\n", best * 1e3,
               (double)n * 16 / (best * 1e-3) / 1e12);
    }
    return 0;
}
