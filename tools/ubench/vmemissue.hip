// What does a VMEM instruction cost the wave that issues it -- and WHY?  (gfx950; round 6)
// tools/ubench/storepath.hip (round 5) measured ~127 cycles per buffer_store and ~100 per load, whatever the width, policy, hit or
// miss and the number of waves per CU: a fixed cost of the issuing wave.  This program asks what it is made of.  One wave per SIMD
// (256 workgroups x 4 waves), every wave runs NV independent-ish v_fma per iteration (8 accumulators) plus NS dwordx4 stores or
// NL dwordx4 loads to / from a stream of its own; MODE says how the memory instruction's registers are treated:
//   stores: 0 = no store; 1 = the store's DATA registers are overwritten right behind it (a temporary reused at once: what the
//           compiler does with the packed state rows of the forward sweep); 2 = they are written right BEFORE the store and rest
//           for the whole iteration after it; 3 = as 1, but an s_nop 7 x 2 sits between the store and the overwrite;
//           4 = dwordx4 stores from FOUR register sets used in turn (a set is rewritten 4 stores later)
//   loads:  5 = NL loads back to back at the top; 6 = NL loads spread, one per NV / NL fmas; (data consumed one iteration later)
// Output: cycles per iteration (s_memtime of wave 0 of workgroup 0 .. and the mean over all waves).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/vmemissue.hip -o /tmp/vmemissue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
template <int MODE, int NV, int NM>
__global__ void __launch_bounds__(256) kern(char *stb, const char *ldb, float *out, long long *cyc, int iters, unsigned wrap)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), W = blockDim.x >> 6, b = blockIdx.x;
    __amdgpu_buffer_rsrc_t rs = make_rsrc(stb + ((size_t)b * W + wave) * wrap, wrap);
    __amdgpu_buffer_rsrc_t rl = make_rsrc(ldb + ((size_t)b * W + wave) * wrap, wrap);
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 1.0f + lane * 1e-3f + i;
    const float a = 0.999f, c = 1e-3f;
    u32x4 d[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = (u32x4){(unsigned)lane, 1u, 2u, (unsigned)i};
    u32x4 ld[NM > 0 ? NM : 1];
#pragma unroll
    for (int i = 0; i < NM; ++i) ld[i] = (u32x4){0u, 0u, 0u, 0u};
    unsigned so = 0;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 5 || MODE == 6) {   // consume last iteration's loads
            unsigned x = 0;
#pragma unroll
            for (int i = 0; i < NM; ++i) x ^= ld[i].x ^ ld[i].w;
            acc[0] += (float)(x & 1u) * 1e-9f;
        }
        if constexpr (MODE == 5) {
#pragma unroll
            for (int i = 0; i < NM; ++i) ld[i] = __builtin_amdgcn_raw_buffer_load_b128(rl, lane * 16 + i * 1024, so, 2);
        }
        constexpr int SEG = NM > 0 ? NM : 1;
#pragma unroll
        for (int sgm = 0; sgm < SEG; ++sgm) {
#pragma unroll
            for (int i = 0; i < NV / SEG; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i & 7]) : "v"(a), "v"(c));
            if constexpr (MODE == 6) ld[sgm] = __builtin_amdgcn_raw_buffer_load_b128(rl, lane * 16 + sgm * 1024, so, 2);
            if constexpr (MODE >= 1 && MODE <= 4) {
                constexpr int S = MODE == 4 ? 4 : 1;
                u32x4 &dd = d[sgm % S];
                if constexpr (MODE == 2 || MODE == 4) {   // write the data registers right before the store
                    asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %4\n\tv_mov_b32 %3, %4" : "=v"(dd.x), "=v"(dd.y), "=v"(dd.z), "=v"(dd.w) : "v"(acc[sgm & 7]));
                }
                __builtin_amdgcn_raw_buffer_store_b128(dd, rs, lane * 16 + sgm * 1024, so, 16);
                asm volatile("" : : "v"(dd) : "memory");
                if constexpr (MODE == 3) asm volatile("s_nop 7\n\ts_nop 7");
                if constexpr (MODE == 1 || MODE == 3) {   // ... or right behind it (a temporary reused at once)
                    asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %4\n\tv_mov_b32 %2, %4\n\tv_mov_b32 %3, %4" : "=v"(dd.x), "=v"(dd.y), "=v"(dd.z), "=v"(dd.w) : "v"(acc[sgm & 7]));
                }
            }
        }
        so += NM * 1024u;
        if (so + NM * 1024u > wrap) so = 0;
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    s += (float)(d[0].x ^ d[1].y ^ d[2].z ^ d[3].w);
    out[((size_t)b * W + wave) * 64 + lane] = s;
    if (lane == 0) cyc[b * W + wave] = t1 - t0;
}
static int g_B = 256;
static double g_bare[4096];   // cycles per iteration of the bare chain, by NV
template <int MODE, int NV, int NM>
void run(const char *what, char *stb, char *ldb, float *out, long long *cyc, unsigned wrap)
{
    const int iters = 200, B = g_B, W = 4;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    kern<MODE, NV, NM><<<B, W * 64>>>(stb, ldb, out, cyc, iters, wrap);
    CHECK(hipEventRecord(e0));
    kern<MODE, NV, NM><<<B, W * 64>>>(stb, ldb, out, cyc, iters, wrap);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(B * W);
    CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    double mean = 0; long long mx = 0;
    for (auto v : h) { mean += (double)v; mx = v > mx ? v : mx; }
    mean /= h.size();
    double *bare = g_bare;
    if (MODE == 0) bare[NV] = mean / iters;
    const double gb = (double)B * W * NM * 1024.0 * iters / (ms * 1e-3) / 1e12;
    printf("%-78s NV=%3d NM=%2d  %7.1f us  cycles/iteration mean %7.0f max %7.0f  -> %6.1f cycles per memory instruction over the bare chain; %5.2f TB/s\n", what, NV, NM, ms * 1e3,
           mean / iters, (double)mx / iters, MODE ? (mean / iters - bare[NV]) / NM : 0.0, MODE ? gb : 0.0);
}
int main()
{
    const unsigned wrap = 4u << 20;   // 4 MB per wave: streams, 4 GB in total
    char *stb, *ldb; float *out; long long *cyc;
    CHECK(hipMalloc(&stb, (size_t)4096 * wrap)); CHECK(hipMalloc(&ldb, (size_t)4096 * wrap));
    CHECK(hipMalloc(&out, 4096 * 64 * 4)); CHECK(hipMalloc(&cyc, 4096 * 8));
    CHECK(hipMemset(ldb, 0, (size_t)4096 * wrap));
  // (passes 3, 4: two and four workgroups per CU -- 8 and 16 waves per CU: does the memory system give more to more waves?)
  for (int pass = 0; pass < 5; ++pass) {
    g_B = pass == 0 ? 256 : (pass == 1 ? 64 : (pass == 2 ? 16 : (pass == 3 ? 512 : 1024)));
    printf("== %d workgroups x 4 waves (one per SIMD and workgroup); per iteration NV v_fma (8 independent accumulators) + NM memory instructions (dwordx4, 1 KB each)\n", g_B);
    run<0, 320, 10>("no memory instruction (the bare chain)", stb, ldb, out, cyc, wrap);
    run<1, 320, 10>("10 stores, data registers OVERWRITTEN right behind each store", stb, ldb, out, cyc, wrap);
    run<2, 320, 10>("10 stores, data registers written right BEFORE each store, at rest behind it", stb, ldb, out, cyc, wrap);
    run<3, 320, 10>("10 stores, overwritten behind the store after s_nop 7 x 2", stb, ldb, out, cyc, wrap);
    run<4, 320, 10>("10 stores from four register sets in turn (written before the store)", stb, ldb, out, cyc, wrap);
    run<5, 320, 16>("16 loads back to back at the top of the iteration", stb, ldb, out, cyc, wrap);
    run<6, 320, 16>("16 loads spread, one per 20 fmas", stb, ldb, out, cyc, wrap);
    run<0, 640, 10>("no memory instruction (the bare chain)", stb, ldb, out, cyc, wrap);
    run<1, 640, 10>("10 stores, data registers OVERWRITTEN right behind each store", stb, ldb, out, cyc, wrap);
    run<2, 640, 10>("10 stores, data registers written right BEFORE each store, at rest behind it", stb, ldb, out, cyc, wrap);
    run<6, 640, 16>("16 loads spread, one per 40 fmas", stb, ldb, out, cyc, wrap);
    run<0, 1920, 10>("no memory instruction (the bare chain)", stb, ldb, out, cyc, wrap);
    run<1, 1920, 10>("10 stores, data registers OVERWRITTEN right behind each store", stb, ldb, out, cyc, wrap);
    run<2, 1920, 10>("10 stores, data registers written right BEFORE each store, at rest behind it", stb, ldb, out, cyc, wrap);
    run<5, 1920, 16>("16 loads back to back at the top of the iteration", stb, ldb, out, cyc, wrap);
    run<6, 1920, 16>("16 loads spread, one per 120 fmas", stb, ldb, out, cyc, wrap);
  }
    return 0;
}
