// What do a sweep's stores cost the waves of a CU?  (gfx950; round 5)
// Both sweeps run a quarter faster in builds that issue no store, whichever wave of the CU issues them and whether they are 8 or
// 16 instructions per chunk.  This program takes the sweeps apart: every wave runs a DEPENDENT chain of v_fma (the recurrence:
// nothing in it can overlap) of ~CH cycles per "chunk", and per chunk
//   * stores ST_KB KB (dwordx4: 1 KB per instruction, or dwordx2: 512 B) to a stream of its own, policy AUXS, either bunched at
//     the end of the chunk or spread through the chain;
//   * optionally loads LD_KB KB of another stream of its own one chunk ahead (nt), folded into the chain when it arrives.
// 256 workgroups x W waves (one workgroup per CU).  WHO = 0: every wave stores its own; 1: wave 0 issues the stores of all W
// waves and runs no chain (a "flusher"); 2: as 1, but wave 0 also runs the chain.  HIT = 1: the store stream wraps inside 64 KB
// per wave (cache-served), the loads inside 64 KB as well.
// Output: cycles per chunk of the slowest wave (s_memtime) and the launch time.
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/storepath.hip -o /tmp/storepath
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

// NV: v_fma per chunk (a dependent chain: ~5 cycles each); NS: store instructions per chunk and wave; X2: dwordx2 instead of dwordx4;
// NL: dwordx4 loads per chunk and wave; SPREAD: a store after every NV / NS links of the chain instead of all at its end
template <int NV, int NS, bool X2, int NL, bool SPREAD, int AUXS, int WHO, int NLDS = 0>
__global__ void __launch_bounds__(512) kern(char *stb, const char *ldb, float *out, long long *cyc, int chunks, unsigned st_wrap, unsigned ld_wrap)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6, b = blockIdx.x;
    constexpr unsigned SB = X2 ? 512u : 1024u;          // bytes per store instruction
    const size_t per_wave_st = (size_t)st_wrap, per_wave_ld = (size_t)ld_wrap;
    __amdgpu_buffer_rsrc_t rs = make_rsrc(stb + ((size_t)b * W + wave) * per_wave_st, (unsigned)per_wave_st);
    __amdgpu_buffer_rsrc_t rl = make_rsrc(ldb + ((size_t)b * W + wave) * per_wave_ld, (unsigned)per_wave_ld);
    float x = 1.0f + lane * 1e-3f;
    const float a = 0.999f, c = 1e-3f;
    u32x4 ld[NL > 0 ? NL : 1];
    unsigned so = 0, lo = 0;
#pragma unroll
    for (int i = 0; i < NL; ++i) ld[i] = __builtin_amdgcn_raw_buffer_load_b128(rl, lane * 16 + i * 1024, 0, 2);
    // WHO = 3: waves W/2 .. W-1 are "flushers" -- flusher w + W/2 (same SIMD as wave w) issues wave w's stores and runs no chain
    const bool chain = WHO == 3 ? wave < W / 2 : !(WHO == 1 && wave == 0);
    const int my_ns = WHO == 0 ? NS : (WHO == 3 ? (wave >= W / 2 ? NS : 0) : (wave == 0 ? NS * (WHO == 1 ? W - 1 : W) : 0));   // stores this wave issues per chunk
    __shared__ float lds[8 * 64 * 33];
    float *my = lds + wave * 64 * 33 + lane * 33;
    const long long t0 = __builtin_readcyclecounter();
    for (int ch = 0; ch < chunks; ++ch) {
        // the loaded data enters the chain (a wait for loads issued a chunk ago)
        if constexpr (NL > 0) {
            unsigned acc = 0;
#pragma unroll
            for (int i = 0; i < NL; ++i) acc ^= ld[i].x ^ ld[i].w;
            x += (float)(acc & 1u) * 1e-9f;
            lo += NL * 1024u;
            if (lo + NL * 1024u > ld_wrap) lo = 0;
#pragma unroll
            for (int i = 0; i < NL; ++i) ld[i] = __builtin_amdgcn_raw_buffer_load_b128(rl, lane * 16 + i * 1024, lo, 2);
        }
        auto store1 = [&](int k) {
            const unsigned v = __float_as_uint(x) + k;
            if constexpr (X2) __builtin_amdgcn_raw_buffer_store_b64((u32x2){v, v ^ 1u}, rs, lane * 8, so, AUXS);
            else __builtin_amdgcn_raw_buffer_store_b128((u32x4){v, v ^ 1u, v ^ 2u, v ^ 3u}, rs, lane * 16, so, AUXS);
            so += SB;
            if (so + SB > st_wrap) so = 0;
        };
        if (WHO == 0) {
            constexpr int SEG = NS > 0 ? NS : 1;
#pragma unroll
            for (int sgm = 0; sgm < SEG; ++sgm) {
#pragma unroll
                for (int i = 0; i < NV / SEG; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(c));
                if constexpr (SPREAD && NS > 0) store1(sgm);
            }
            if constexpr (!SPREAD) {
#pragma unroll
                for (int k = 0; k < NS; ++k) store1(k);
            }
        } else {
            if (chain) {
                constexpr int SEG = NLDS > 0 ? NLDS : 1;
#pragma unroll
                for (int sgm = 0; sgm < SEG; ++sgm) {
#pragma unroll
                    for (int i = 0; i < NV / SEG; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(c));
                    if constexpr (NLDS > 0) my[sgm & 31] = x;   // the sweeps write one output per step into their LDS ring
                }
            }
            for (int k = 0; k < my_ns; ++k) store1(k);
            if (!chain) __builtin_amdgcn_s_sleep(WHO == 3 ? 16 : 64);   // (a flusher that is ahead idles)
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    out[(size_t)b * blockDim.x + threadIdx.x] = x;
    if (lane == 0) cyc[b * W + wave] = t1 - t0;
}

template <int NV, int NS, bool X2, int NL, bool SPREAD, int AUXS, int WHO, int NLDS = 0>
static void run(const char *what, int W, bool hit, char *stb, char *ldb, float *out, long long *cyc, int chunks)
{
    const unsigned stream = 16u << 20;   // per wave: 16 MB of store stream, 16 MB of load stream
    const unsigned st_wrap = hit ? 65536u : stream, ld_wrap = hit ? 65536u : stream;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {   // (first: warm)
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((kern<NV, NS, X2, NL, SPREAD, AUXS, WHO, NLDS>), dim3(256), dim3(64 * W), 0, 0, stb, ldb, out, cyc, chunks, st_wrap, ld_wrap);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
    }
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(256 * W);
    CHECK(hipMemcpy(h.data(), cyc, sizeof(long long) * 256 * W, hipMemcpyDeviceToHost));
    long long mx = 0; double mean = 0;
    for (auto v : h) { mx = v > mx ? v : mx; mean += (double)v; }
    mean /= (double)h.size();
    const double st_bytes = (double)256 * W * chunks * NS * (X2 ? 512.0 : 1024.0) * (WHO == 1 ? (double)(W - 1) / W : (WHO == 3 ? 0.5 : 1.0));
    const double ld_bytes = (double)256 * W * chunks * NL * 1024.0;
    printf("%-58s W=%d %s  %8.1f us  cycles/chunk mean %7.0f max %7.0f   stores %6.2f TB/s  loads %6.2f TB/s\n", what, W, hit ? "hit " : "miss", ms * 1e3,
           mean / chunks, (double)mx / chunks, st_bytes / (ms * 1e-3) / 1e12, ld_bytes / (ms * 1e-3) / 1e12);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

int main()
{
    const size_t total = (size_t)256 * 8 * (16u << 20);   // up to 8 waves per workgroup, 16 MB of stream per wave: 32 GB per buffer
    char *stb, *ldb; float *out; long long *cyc;
    CHECK(hipMalloc(&stb, total)); CHECK(hipMalloc(&ldb, total));
    CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&cyc, 256 * 8 * 8));
    CHECK(hipMemset(ldb, 1, total));
    const int CH = 200;   // chunks per wave: 200 x 10 KB = 2 MB of stores per wave (inside its 16 MB)
    constexpr int NV = 800;
    printf("chain of %d dependent v_fma per chunk; 256 workgroups\n", NV);
    for (int hit = 0; hit < 2; ++hit) {
        const bool h = hit != 0;
        run<NV, 0, false, 0, false, 0, 0>("chain only", 4, h, stb, ldb, out, cyc, CH);
        run<NV, 10, false, 0, false, 16, 0>("+ 10 x dwordx4 stores sc1 at the end (fwd-like)", 4, h, stb, ldb, out, cyc, CH);
        run<NV, 10, false, 0, true, 16, 0>("+ 10 x dwordx4 stores sc1 spread", 4, h, stb, ldb, out, cyc, CH);
        run<NV, 10, false, 0, false, 0, 0>("+ 10 x dwordx4 stores default policy", 4, h, stb, ldb, out, cyc, CH);
        run<NV, 10, false, 0, false, 2, 0>("+ 10 x dwordx4 stores nt", 4, h, stb, ldb, out, cyc, CH);
        run<NV, 10, false, 0, false, 18, 0>("+ 10 x dwordx4 stores nt sc1", 4, h, stb, ldb, out, cyc, CH);
        run<NV, 16, true, 0, false, 18, 0>("+ 16 x dwordx2 stores nt sc1 (bwd-like, 8 KB)", 4, h, stb, ldb, out, cyc, CH);
        run<NV, 8, false, 0, false, 18, 0>("+  8 x dwordx4 stores nt sc1 (8 KB)", 4, h, stb, ldb, out, cyc, CH);
        run<NV, 20, false, 0, false, 16, 0>("+ 20 x dwordx4 stores sc1 (twice the bytes)", 4, h, stb, ldb, out, cyc, CH);
        run<NV, 0, false, 10, false, 0, 0>("+ 10 x dwordx4 loads nt, no stores", 4, h, stb, ldb, out, cyc, CH);
        run<NV, 8, false, 10, false, 18, 0>("+ 10 loads + 8 x dwordx4 stores nt sc1 (bwd-like)", 4, h, stb, ldb, out, cyc, CH);
        run<NV, 10, false, 16, false, 16, 0>("+ 16 loads + 10 x dwordx4 stores sc1 (fwd-like)", 4, h, stb, ldb, out, cyc, CH);
        run<NV, 10, false, 0, false, 16, 1>("flusher: wave 0 issues the other 3 waves' stores, no chain", 4, h, stb, ldb, out, cyc, CH);
        run<NV, 10, false, 0, false, 16, 2>("wave 0 issues all 4 waves' stores AND runs the chain", 4, h, stb, ldb, out, cyc, CH);
        run<NV, 10, false, 0, false, 16, 0>("+ 10 x dwordx4 stores sc1, 1 wave per CU", 1, h, stb, ldb, out, cyc, CH);
        run<NV, 10, false, 0, false, 16, 0>("+ 10 x dwordx4 stores sc1, 2 waves per CU", 2, h, stb, ldb, out, cyc, CH);
        run<NV, 10, false, 0, false, 16, 0>("+ 10 x dwordx4 stores sc1, 8 waves per CU", 8, h, stb, ldb, out, cyc, CH);
        run<NV, 0, false, 0, false, 0, 0>("chain only, 8 waves per CU", 8, h, stb, ldb, out, cyc, CH);
        // a flusher wave beside every chain wave, on its SIMD (8 waves: 0-3 chain, 4-7 flushers issuing 16 x dwordx2 each per chunk)
        run<NV, 0, true, 0, false, 18, 3>("4 chain waves + 4 idle flushers (no stores)", 8, h, stb, ldb, out, cyc, CH);
        run<NV, 16, true, 0, false, 18, 3>("4 chain waves + 4 flushers x 16 dwordx2 stores", 8, h, stb, ldb, out, cyc, CH);
        run<NV, 0, true, 0, false, 18, 3, 32>("4 chain waves with 32 LDS writes + 4 idle flushers", 8, h, stb, ldb, out, cyc, CH);
        run<NV, 16, true, 0, false, 18, 3, 32>("4 chain waves with 32 LDS writes + 4 flushers x 16 stores", 8, h, stb, ldb, out, cyc, CH);
    }
    return 0;
}
