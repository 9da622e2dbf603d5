// Does the WIDTH of the float2 state accesses matter once the memory system pushes back?  (gfx950; round 6)
// The adjoint forward sweep moves, per wave and 16-step chunk: 16 rows of Q in (dwordx2 per lane, 512 B per instruction), one
// staged input block in (4 x dwordx4), 16 rows of Qd out (dwordx2, one per step) -- and runs at 5.0 TB/s of 1.5 GB where its
// instruction stream alone (everything cache-served) needs 60 % of the time.  This program runs the same byte mix per iteration
// from every wave of a full chip (256 workgroups x 4 waves, a stream of its own per wave and tensor), between NV fmas, as
//   WIDE = 0: 16 dwordx2 loads (back to back) + 4 dwordx4 loads + 16 dwordx2 stores spread over the fmas
//   WIDE = 1:  8 dwordx4 loads               + 4 dwordx4 loads +  8 dwordx4 stores (two steps adjacent per lane)
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/f2mix.hip -o /tmp/f2mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}
template <int WIDE, int NV>
__global__ void __launch_bounds__(256) kern(char *qd, const char *q, const char *z, float *out, int iters, unsigned wrap, unsigned skew)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), W = blockDim.x >> 6, b = blockIdx.x;
    const size_t w = (size_t)b * W + wave;
    // (skew: every wave's streams start `skew` bytes further than a multiple of the 4 MB stride -- do power-of-two strides camp on channels?)
    __amdgpu_buffer_rsrc_t rs = make_rsrc(qd + w * (wrap + skew), wrap), rq = make_rsrc(q + w * (wrap + skew), wrap), rz = make_rsrc(z + w * (wrap / 2 + skew), wrap / 2);
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 1.0f + lane * 1e-3f + i;
    const float a = 0.999f, c = 1e-3f;
    u32x2 l2[16]; u32x4 l4[8], lz[4];
#pragma unroll
    for (int i = 0; i < 16; ++i) l2[i] = (u32x2){0u, 0u};
#pragma unroll
    for (int i = 0; i < 8; ++i) l4[i] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < 4; ++i) lz[i] = (u32x4){0u, 0u, 0u, 0u};
    unsigned so = 0, zo = 0;
    for (int it = 0; it < iters; ++it) {
        unsigned x = 0;   // consume last iteration's loads
        if constexpr (WIDE) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x ^= l4[i].x ^ l4[i].w;
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) x ^= l2[i].x ^ l2[i].y;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) x ^= lz[i].y;
        acc[0] += (float)(x & 1u) * 1e-9f;
        if constexpr (WIDE) {
#pragma unroll
            for (int i = 0; i < 8; ++i) l4[i] = __builtin_amdgcn_raw_buffer_load_b128(rq, lane * 16 + i * 1024, so, 2);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) l2[i] = __builtin_amdgcn_raw_buffer_load_b64(rq, lane * 8 + i * 512, so, 2);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) lz[i] = __builtin_amdgcn_raw_buffer_load_b128(rz, lane * 16 + i * 1024, zo, 0);
#pragma unroll
        for (int sgm = 0; sgm < 16; ++sgm) {
#pragma unroll
            for (int i = 0; i < NV / 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i & 7]) : "v"(a), "v"(c));
            if constexpr (WIDE) {
                if (sgm & 1) {
                    u32x4 d = (u32x4){__float_as_uint(acc[sgm & 7]), __float_as_uint(acc[(sgm + 1) & 7]), __float_as_uint(acc[(sgm + 2) & 7]), (unsigned)sgm};
                    __builtin_amdgcn_raw_buffer_store_b128(d, rs, lane * 16 + (sgm >> 1) * 1024, so, 16);
                }
            } else {
                u32x2 d = (u32x2){__float_as_uint(acc[sgm & 7]), __float_as_uint(acc[(sgm + 1) & 7])};
                __builtin_amdgcn_raw_buffer_store_b64(d, rs, lane * 8 + sgm * 512, so, 16);
            }
        }
        so += 8192u; zo += 4096u;
        if (so + 8192u > wrap) so = 0, zo = 0;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[w * 64 + lane] = s;
}
static unsigned g_skew = 0;
template <int WIDE, int NV>
void run(const char *what, int B, char *qd, char *q, char *z, float *out, unsigned wrap)
{
    const int iters = 288, W = 4;   // (a sweep's 72 chunks x 4)
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        kern<WIDE, NV><<<B, W * 64>>>(qd, q, z, out, iters, wrap, g_skew);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    const double gb = (double)B * W * (8192.0 * 2 + 4096.0) * iters / (best * 1e-3) / 1e12;
    printf("%-60s B=%3d NV=%4d  %7.1f us  %5.2f TB/s\n", what, B, NV, best * 1e3, gb);
}
int main(int argc, char **argv)
{
    const unsigned wrap = 4u << 20;
    g_skew = argc > 1 ? (unsigned)atoi(argv[1]) : 0u;   // bytes; the allocations leave room for 64 KB per wave
    printf("skew %u bytes per wave\n", g_skew);
    char *qd, *q, *z; float *out;
    CHECK(hipMalloc(&qd, (size_t)1024 * (wrap + 65536))); CHECK(hipMalloc(&q, (size_t)1024 * (wrap + 65536))); CHECK(hipMalloc(&z, (size_t)1024 * (wrap / 2 + 65536)));
    CHECK(hipMalloc(&out, 1024 * 64 * 4));
    CHECK(hipMemset(q, 0, (size_t)1024 * wrap)); CHECK(hipMemset(z, 0, (size_t)1024 * wrap / 2));
    for (int B : {256, 64}) {
        run<0, 160>("dwordx2 rows, little arithmetic", B, qd, q, z, out, wrap);
        run<1, 160>("dwordx4 rows, little arithmetic", B, qd, q, z, out, wrap);
        run<0, 640>("dwordx2 rows, 640 fmas per chunk (~2600 cycles)", B, qd, q, z, out, wrap);
        run<1, 640>("dwordx4 rows, 640 fmas per chunk", B, qd, q, z, out, wrap);
        run<0, 1280>("dwordx2 rows, 1280 fmas per chunk (~5100 cycles)", B, qd, q, z, out, wrap);
        run<1, 1280>("dwordx4 rows, 1280 fmas per chunk", B, qd, q, z, out, wrap);
    }
    return 0;
}
