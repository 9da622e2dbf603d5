// Memory-system microbenchmark for the traffic MIX of the forward sweep (gfx950): per "chunk" a wave loads 16 KB
// (16 x buffer_load_dwordx4, prefetched DEPTH chunks ahead in registers) and stores OUT_BYTES (9 KB) of state, as
// dwordx3 (16 instr, 12-byte lane stride -- what the packed state does), dwordx4 (12 instr) or dwordx2 (24 instr),
// with a chosen cache policy, optionally with a dependent VALU chain per chunk that stands in for the recurrence.
// Total traffic is that of B=256, N=M=512: 537 MB read + 32768 x OUT_BYTES written (302 MB at 9 KB), whatever the geometry.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/mix.hip -o tools/ubench/mix
// (round 5: the record the forward sweep writes per wave and 32-step chunk is 9216 B -- the 18-bit packed state; -DOUT_BYTES=10240: the 20-bit
//  state, 12288: the 24-bit state of rounds 1-3)
#ifndef OUT_BYTES
#define OUT_BYTES 9216
#endif
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

// STW: dwords per lane per store; DEPTH: chunks of loads in flight; AUXS/AUXL: cache policy of stores / loads
// RD / WR: do loads / stores at all
template <int STW, int DEPTH, int AUXS, int AUXL, bool RD, bool WR, int MAXT>
__global__ void __launch_bounds__(MAXT) mix(const char *in, char *out, int chunks, int work)
{
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int wave = blockIdx.x * wpb + (threadIdx.x >> 6);
    const size_t in_bytes = (size_t)chunks * 16384, out_bytes = (size_t)chunks * OUT_BYTES;
    __amdgpu_buffer_rsrc_t ri = make_rsrc(in + (size_t)wave * in_bytes, (unsigned)in_bytes);
    __amdgpu_buffer_rsrc_t ro = make_rsrc(out + (size_t)wave * out_bytes, (unsigned)out_bytes);
    u32x4 ring[DEPTH][16];
    if (RD) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int i = 0; i < 16; ++i) ring[d][i] = __builtin_amdgcn_raw_buffer_load_b128(ri, lane * 16 + i * 1024, d * 16384, AUXL);
    }
    float acc = (float)lane;
    for (int c0 = 0; c0 < chunks; c0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int c = c0 + d;
            unsigned s = 0;
            if (RD) {
                const int cn = c + DEPTH < chunks ? c + DEPTH : c;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    s += ring[d][i][0] ^ ring[d][i][1] ^ ring[d][i][2] ^ ring[d][i][3];
                    ring[d][i] = __builtin_amdgcn_raw_buffer_load_b128(ri, lane * 16 + i * 1024, cn * 16384, AUXL);
                }
            }
            for (int w = 0; w < work; ++w) acc = __builtin_fmaf(acc, 1.0000001f, 1e-9f);  // dependent chain, ~5 cycles each
            s += __float_as_uint(acc);
            if (WR) {
                constexpr int NST = OUT_BYTES / (64 * STW * 4);
#pragma unroll
                for (int i = 0; i < NST; ++i) {
                    if constexpr (STW == 4) {
                        u32x4 v = {s, s + 1, s + 2, s + 3};
                        __builtin_amdgcn_raw_buffer_store_b128(v, ro, lane * 16 + i * 1024, c * OUT_BYTES, AUXS);
                    } else if constexpr (STW == 3) {
                        u32x3 v = {s, s + 1, s + 2};
                        __builtin_amdgcn_raw_buffer_store_b96(v, ro, lane * 12 + i * 768, c * OUT_BYTES, AUXS);
                    } else {
                        u32x2 v = {s, s + 1};
                        __builtin_amdgcn_raw_buffer_store_b64(v, ro, lane * 8 + i * 512, c * OUT_BYTES, AUXS);
                    }
                }
            }
        }
    }
    if (acc == 12345.f) out[0] = 1;
}

int main(int argc, char **argv)
{
    const size_t total_chunks = 256 * 4 * 32;   // 32768 chunks: 537 MB in, 302 MB out at 9 KB per chunk (the state without its padding)
    char *in, *out;
    CHECK(hipMalloc(&in, total_chunks * 16384 + 65536));
    CHECK(hipMalloc(&out, total_chunks * OUT_BYTES + 65536));
    CHECK(hipMemset(in, 1, total_chunks * 16384));
    CHECK(hipMemset(out, 0, total_chunks * OUT_BYTES));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto bench = [&](const char *name, auto kern, int blocks, int threads, int work, bool rd, bool wr) {
        const int waves = blocks * threads / 64;
        const int chunks = (int)(total_chunks / waves);
        float best = 1e9;
        for (int pass = 0; pass < 3; ++pass) {
            CHECK(hipEventRecord(e0));
            for (int it = 0; it < 4; ++it) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, in, out, chunks, work);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            best = ms / 4 < best ? ms / 4 : best;
        }
        const double bytes = (double)total_chunks * ((rd ? 16384 : 0) + (wr ? OUT_BYTES : 0));
        printf("%-44s grid %4d x %3d work %4d: %7.1f us  %5.2f TB/s\n", name, blocks, threads, work, best * 1e3, bytes / (best * 1e-3) / 1e12);
        fflush(stdout);
    };
#define B3(NAME, STW, DEPTH, AUXS, AUXL, RD, WR, WORK)                                                \
    bench(NAME, mix<STW, DEPTH, AUXS, AUXL, RD, WR, 256>, 256, 256, WORK, RD, WR);                     \
    bench(NAME, mix<STW, DEPTH, AUXS, AUXL, RD, WR, 256>, 512, 256, WORK, RD, WR);                     \
    bench(NAME, mix<STW, DEPTH, AUXS, AUXL, RD, WR, 512>, 256, 512, WORK, RD, WR);                     \
    bench(NAME, mix<STW, DEPTH, AUXS, AUXL, RD, WR, 256>, 1024, 256, WORK, RD, WR);
    // read only / write only
    B3("read only  x4 D1 nt", 4, 1, 0, 2, true, false, 0)
    B3("read only  x4 D2 nt", 4, 2, 0, 2, true, false, 0)
    B3("write only x3 default", 3, 1, 0, 0, false, true, 0)
    B3("write only x4 default", 4, 1, 0, 0, false, true, 0)
    B3("write only x2 default", 2, 1, 0, 0, false, true, 0)
    B3("write only x3 nt", 3, 1, 2, 0, false, true, 0)
    B3("write only x4 nt", 4, 1, 2, 0, false, true, 0)
    B3("write only x4 sc1", 4, 1, 16, 0, false, true, 0)
    B3("write only x4 sc0sc1", 4, 1, 17, 0, false, true, 0)
    // the mix, no compute
    B3("mix x3 D1 st default, ld nt", 3, 1, 0, 2, true, true, 0)
    B3("mix x4 D1 st default, ld nt", 4, 1, 0, 2, true, true, 0)
    B3("mix x2 D1 st default, ld nt", 2, 1, 0, 2, true, true, 0)
    B3("mix x3 D2 st default, ld nt", 3, 2, 0, 2, true, true, 0)
    B3("mix x4 D2 st default, ld nt", 4, 2, 0, 2, true, true, 0)
    B3("mix x3 D1 st nt, ld nt", 3, 1, 2, 2, true, true, 0)
    B3("mix x4 D1 st nt, ld nt", 4, 1, 2, 2, true, true, 0)
    B3("mix x4 D1 st sc1, ld nt", 4, 1, 16, 2, true, true, 0)
    B3("mix x4 D1 st default, ld default", 4, 1, 0, 0, true, true, 0)
    // the mix with a dependent chain per chunk (32 steps x ~33 instructions ~ 1000 issue slots)
    B3("mix x3 D1 + chain", 3, 1, 0, 2, true, true, 1000)
    B3("mix x4 D1 + chain", 4, 1, 0, 2, true, true, 1000)
    B3("mix x3 D2 + chain", 3, 2, 0, 2, true, true, 1000)
    B3("mix x4 D2 + chain", 4, 2, 0, 2, true, true, 1000)
    B3("chain only", 4, 1, 0, 2, false, false, 1000)
    return 0;
}
