// Does the ORDER in which the chip sweeps memory matter for a read/write mix?  (gfx950)
// Every wave moves `chunks` chunks: loads RD_KB, stores WR_KB per chunk (16-byte accesses, 1 KB per wave instruction).
//   layout 0: each wave owns one contiguous region (what the soft-DP state layout [pair][strip][t][lane] does)
//   layout 1: chunk c of wave w lives at (c * nwaves + w): all waves march through memory together
//   layout 2: as 1 but the wave index is permuted so that the waves of one workgroup are far apart
// plus a plain grid-stride float4 copy at full occupancy as the reference for "what the chip can do".
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mix2.hip -o tools/ubench/mix2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int RD_KB, int WR_KB, int LAYOUT, int AUXS, int AUXL>
__global__ void __launch_bounds__(256) mixk(const u32x4 *in, u32x4 *out, int chunks, int nwaves, unsigned *sink)
{
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    u32x4 ring[RD_KB > 0 ? RD_KB : 1];
    auto in_at = [&](int c) -> const u32x4 * {
        const size_t idx = LAYOUT == 0 ? (size_t)wave * chunks + c : (size_t)c * nwaves + wave;
        return in + idx * (RD_KB * 64) + lane;
    };
    auto out_at = [&](int c) -> u32x4 * {
        const size_t idx = LAYOUT == 0 ? (size_t)wave * chunks + c : (size_t)c * nwaves + wave;
        return out + idx * (WR_KB * 64) + lane;
    };
    if (RD_KB > 0) {
        const u32x4 *p = in_at(0);
#pragma unroll
        for (int i = 0; i < RD_KB; ++i) ring[i] = AUXL ? __builtin_nontemporal_load(p + i * 64) : p[i * 64];
    }
    unsigned s = lane;
    for (int c = 0; c < chunks; ++c) {
        if (RD_KB > 0) {
            const u32x4 *p = in_at(c + 1 < chunks ? c + 1 : c);
#pragma unroll
            for (int i = 0; i < RD_KB; ++i) {
                s += ring[i][0] ^ ring[i][1] ^ ring[i][2] ^ ring[i][3];
                ring[i] = AUXL ? __builtin_nontemporal_load(p + i * 64) : p[i * 64];
            }
        }
        if (WR_KB > 0) {
            u32x4 *q = out_at(c);
#pragma unroll
            for (int i = 0; i < WR_KB; ++i) {
                const u32x4 v = {s, s + i, s ^ 5u, s + 7u};
                if (AUXS) __builtin_nontemporal_store(v, q + i * 64);
                else q[i * 64] = v;
            }
        }
    }
    if (s == 0x12345678u) sink[0] = s;
}

__global__ void __launch_bounds__(256) copy4(const u32x4 *in, u32x4 *out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
__global__ void __launch_bounds__(256) read4(const u32x4 *in, unsigned *sink, size_t n)
{
    unsigned s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const u32x4 v = in[i];
        s += v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (s == 0x12345678u) sink[0] = s;
}
__global__ void __launch_bounds__(256) fill4(u32x4 *out, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = (u32x4){1u, 2u, 3u, (unsigned)i};
}

int main()
{
    const size_t total_chunks = 32768;   // x 16 KB = 537 MB
    const size_t bytes = total_chunks * 16384;
    u32x4 *in, *out;
    unsigned *sink;
    CHECK(hipMalloc(&in, bytes + 65536));
    CHECK(hipMalloc(&out, bytes + 65536));
    CHECK(hipMalloc(&sink, 256));
    CHECK(hipMemset(in, 1, bytes));
    CHECK(hipMemset(out, 0, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto time_it = [&](auto launch) {
        float best = 1e9;
        for (int pass = 0; pass < 3; ++pass) {
            CHECK(hipEventRecord(e0));
            for (int it = 0; it < 4; ++it) launch();
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms / 4 < best ? ms / 4 : best;
        }
        return best;
    };
    for (int blocks : {256, 1024, 2048, 4096}) {
        const size_t n = bytes / 16;
        float t = time_it([&] { hipLaunchKernelGGL(copy4, dim3(blocks), dim3(256), 0, 0, in, out, n); });
        printf("copy4 grid-stride, %4d blocks: %7.1f us  %5.2f TB/s (read + write)\n", blocks, t * 1e3, 2.0 * bytes / (t * 1e-3) / 1e12);
        t = time_it([&] { hipLaunchKernelGGL(read4, dim3(blocks), dim3(256), 0, 0, in, sink, n); });
        printf("read4 grid-stride, %4d blocks: %7.1f us  %5.2f TB/s\n", blocks, t * 1e3, 1.0 * bytes / (t * 1e-3) / 1e12);
        t = time_it([&] { hipLaunchKernelGGL(fill4, dim3(blocks), dim3(256), 0, 0, out, n); });
        printf("fill4 grid-stride, %4d blocks: %7.1f us  %5.2f TB/s\n", blocks, t * 1e3, 1.0 * bytes / (t * 1e-3) / 1e12);
    }
    auto bench = [&](const char *name, auto kern, int rd_kb, int wr_kb) {
        for (int blocks : {256, 512, 1024}) {
            const int nwaves = blocks * 4;
            const int chunks = (int)(total_chunks / nwaves);
            const float t = time_it([&] { hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, in, out, chunks, nwaves, sink); });
            const double b = (double)total_chunks * (rd_kb + wr_kb) * 1024;
            printf("%-34s rd %2d KB wr %2d KB per chunk, grid %4d: %7.1f us  %5.2f TB/s\n", name, rd_kb, wr_kb, blocks, t * 1e3, b / (t * 1e-3) / 1e12);
            fflush(stdout);
        }
    };
    bench("read  per-wave regions", mixk<16, 0, 0, 0, 0>, 16, 0);
    bench("read  marching", mixk<16, 0, 1, 0, 0>, 16, 0);
    bench("read  per-wave regions nt", mixk<16, 0, 0, 0, 1>, 16, 0);
    bench("read  marching nt", mixk<16, 0, 1, 0, 1>, 16, 0);
    bench("write per-wave regions", mixk<0, 12, 0, 0, 0>, 0, 12);
    bench("write marching", mixk<0, 12, 1, 0, 0>, 0, 12);
    bench("write per-wave regions nt", mixk<0, 12, 0, 1, 0>, 0, 12);
    bench("write marching nt", mixk<0, 12, 1, 1, 0>, 0, 12);
    bench("mix   per-wave regions", mixk<16, 12, 0, 0, 0>, 16, 12);
    bench("mix   marching", mixk<16, 12, 1, 0, 0>, 16, 12);
    bench("mix   per-wave regions, ld nt", mixk<16, 12, 0, 0, 1>, 16, 12);
    bench("mix   marching, ld nt", mixk<16, 12, 1, 0, 1>, 16, 12);
    bench("mix   marching, ld nt st nt", mixk<16, 12, 1, 1, 1>, 16, 12);
    bench("mix 16:8 per-wave regions", mixk<16, 8, 0, 0, 1>, 16, 8);
    bench("mix 16:8 marching", mixk<16, 8, 1, 0, 1>, 16, 8);
    bench("mix 12:8 per-wave (bwd-like)", mixk<12, 8, 0, 0, 1>, 12, 8);
    bench("mix 12:8 marching (bwd-like)", mixk<12, 8, 1, 0, 1>, 12, 8);
    return 0;
}
