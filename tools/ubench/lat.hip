// Dependent-chain latency microbenchmarks for gfx950 (one wave, one block).
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/lat.hip -o tools/ubench/lat
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

#define REP 64
#define ITERS 256

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int old, int src) { return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double old, double src)
{
    const int lo = dpp_i32<CTRL>(__double2loint(old), __double2loint(src));
    const int hi = dpp_i32<CTRL>(__double2hiint(old), __double2hiint(src));
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float old, float src)
{
    return __int_as_float(dpp_i32<CTRL>(__float_as_int(old), __float_as_int(src)));
}

template <int OP, int NCH>
__global__ void lat(long long *out, float seed)
{
    double d = seed + threadIdx.x * 1e-3, d2 = 1.0000001, d3 = 0.999;
    float f = seed + threadIdx.x * 1e-3f, f2 = 1.0000001f, f3 = 0.25f;
    // nchain independent chains interleaved (1 = pure latency; more = throughput)
    double da[4] = {d, d + 1, d + 2, d + 3};
    float fa[4] = {f, f + 1, f + 2, f + 3};
    int ia[4] = {(int)seed, (int)seed + 1, (int)seed + 2, (int)seed + 3};
    int i2 = (int)(seed * 3.f) + threadIdx.x, i3 = (int)(seed * 5.f) - threadIdx.x;
    long long t0 = clock64();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REP; ++r) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if constexpr (OP == 0) da[c] = da[c] + d2;                                    // v_add_f64
                if constexpr (OP == 1) da[c] = fmax(da[c], d3);                               // v_max_f64
                if constexpr (OP == 2) da[c] = __builtin_fma(da[c], d2, d3);                  // v_fma_f64
                if constexpr (OP == 3) fa[c] = (float)((double)fa[c] * d2);                   // cvt f32->f64, mul_f64, cvt f64->f32
                if constexpr (OP == 4) fa[c] = __builtin_amdgcn_exp2f(fa[c]);                 // v_exp_f32
                if constexpr (OP == 5) fa[c] = __builtin_amdgcn_logf(fa[c]);                  // v_log_f32
                if constexpr (OP == 6) fa[c] = __builtin_amdgcn_rcpf(fa[c]);                  // v_rcp_f32
                if constexpr (OP == 7) fa[c] = __builtin_fmaf(fa[c], f2, f3);                 // v_fma_f32
                if constexpr (OP == 8) fa[c] = dpp_f32<0x138>(f3, fa[c]);                     // dpp wave_shr f32
                if constexpr (OP == 9) da[c] = dpp_f64<0x138>(d3, da[c]);                     // dpp wave_shr f64 (2 movs)
                if constexpr (OP == 10) fa[c] = __builtin_fmaf(dpp_f32<0x138>(f3, fa[c]), f2, f3);  // dpp + fma
                if constexpr (OP == 11) fa[c] = __builtin_amdgcn_ldexpf(fa[c], 1);            // v_ldexp_f32
                if constexpr (OP == 12) fa[c] = fa[c] + f2;                                   // v_add_f32
                if constexpr (OP == 13) fa[c] = __shfl_up(fa[c], 1);                          // ds_bpermute path
                if constexpr (OP == 14) fa[c] = dpp_f32<0x111>(f3, fa[c]);                    // dpp row_shr:1
                if constexpr (OP == 15) da[c] = dpp_f64<0x138>(d3, da[c]) + d2;               // dpp f64 + add_f64
                if constexpr (OP == 16) fa[c] = __builtin_amdgcn_ldexpf(fa[c], ia[c] & 3) + f3;         // ldexp + add
                if constexpr (OP == 17) fa[c] = __builtin_amdgcn_frexp_mantf(fa[c]) + f2;              // frexp_mant + add
                if constexpr (OP == 18) { ia[c] = __builtin_amdgcn_frexp_expf(fa[c]) + ia[c]; fa[c] = __int_as_float((ia[c] & 0xff) | 0x3f800000); }  // frexp_exp + add + and_or
                if constexpr (OP == 19) ia[c] = max(max(ia[c], i2), i3) + 1;                             // max3_i32 + add
                if constexpr (OP == 20) ia[c] = ia[c] + i2 + i3;                                        // add3
                if constexpr (OP == 21) fa[c] = __builtin_floorf(fa[c] * f2) + f3;                      // mul + floor + add
                if constexpr (OP == 22) { ia[c] = (int)fa[c]; fa[c] = (float)ia[c] * f2; }               // cvt_i32_f32 + cvt_f32_i32 + mul
            }
        }
    }
    long long t1 = clock64();
    double acc = 0;
    for (int c = 0; c < 4; ++c) acc += da[c] + fa[c] + ia[c];
    if (threadIdx.x == 0) out[0] = t1 - t0;
    if (acc == 12345.678) out[1] = 1;
}

template <int OP, int NCH>
void run1(const char *name, long long *d)
{
    long long h[2];
    hipLaunchKernelGGL((lat<OP, NCH>), dim3(1), dim3(64), 0, 0, d, 1.5f);
    hipLaunchKernelGGL((lat<OP, NCH>), dim3(1), dim3(64), 0, 0, d, 1.5f);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-34s chains=%d  %.2f ticks/op\n", name, NCH, (double)h[0] / ((double)ITERS * REP * NCH));
}
template <int OP>
void run(const char *name, long long *d)
{
    run1<OP, 1>(name, d);
    run1<OP, 2>(name, d);
    run1<OP, 4>(name, d);
}

int main()
{
    long long *d;
    (void)hipMalloc(&d, 64);
    // clock64 tick rate vs shader clock is reported by the f32 fma line (known ~4-5 cycles dependent)
    run<7>("v_fma_f32", d);
    run<12>("v_add_f32", d);
    run<0>("v_add_f64", d);
    run<1>("v_max_f64", d);
    run<2>("v_fma_f64", d);
    run<3>("cvt_f64_f32+mul_f64+cvt_f32_f64", d);
    run<4>("v_exp_f32", d);
    run<5>("v_log_f32", d);
    run<6>("v_rcp_f32", d);
    run<11>("v_ldexp_f32", d);
    run<8>("dpp wave_shr f32", d);
    run<14>("dpp row_shr f32", d);
    run<9>("dpp wave_shr f64 (2 movs)", d);
    run<10>("dpp wave_shr f32 + fma", d);
    run<15>("dpp wave_shr f64 + add_f64", d);
    run<13>("__shfl_up (bpermute)", d);
    run<16>("v_ldexp_f32 + add", d);
    run<17>("v_frexp_mant_f32 + add", d);
    run<18>("v_frexp_exp_i32 + add + and_or", d);
    run<19>("v_max3_i32 + add", d);
    run<20>("v_add3_u32", d);
    run<21>("mul + v_floor_f32 + add", d);
    run<22>("cvt_i32_f32 + cvt_f32_i32 + mul", d);
    return 0;
}
