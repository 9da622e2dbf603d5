// cost of the forward sweep's per-step dependency chain in isolation (no memory traffic)
// usage: chain [waves_per_wg]   -- prints ns/step for a few variants of the body
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int CTRL>
__device__ __forceinline__ float dpp_f(float old, float src)
{
    return __uint_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, 0xf, false));
}

// MODE bits: 1 = exp2 of inputs inside the loop, 2 = rcp + 3 muls for q, 4 = dpp (else plain), 8 = range tracking,
// 16 = slow chain (exponent alignment + frexp) instead of the windowed chain
template <int MODE>
__global__ void __launch_bounds__(512) chain(float *out, const float *in, int steps)
{
    const int lane = threadIdx.x & 63;
    float x = 0.5f + lane * 1e-3f, d = 0.25f, sc = 1.0f + in[0];
    int xe = 1, de = 1;
    unsigned mx = 0, mn = ~0u;
    float acc = 0.f;
    const float t0 = in[1 + (lane & 15)], a0 = in[17 + (lane & 15)];
    for (int s = 0; s < steps; s += 16) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float ct = t0 + k * 1e-3f, ca = a0 - k * 1e-3f;
            if (MODE & 1) {
                ct = __builtin_amdgcn_exp2f(ct * 1.44269504f);
                ca = __builtin_amdgcn_exp2f(ca * 1.44269504f);
            }
            if (!(MODE & 16)) {
                const float ua = (MODE & 4) ? dpp_f<0x138>(d, x) : x + 1e-9f;
                const float u = ua * sc;
                const float ssum = __builtin_fmaf(ca, u + x, d);
                if (MODE & 2) {
                    const float tq = ca * __builtin_amdgcn_rcpf(ssum);
                    acc += tq * u + tq * x;
                }
                d = u;
                x = ct * ssum;
                if (MODE & 8) {
                    mx = max(max(mx, __float_as_uint(u)), __float_as_uint(x));
                    mn = min(mn, __float_as_uint(x));
                }
                x = __builtin_amdgcn_frexp_mantf(x) * 0.0f + x * 0.25f;  // keep it bounded (2 extra ops, both variants... )
            } else {
                const float ua = (MODE & 4) ? dpp_f<0x138>(d, x) : x + 1e-9f;
                const int ue = (MODE & 4) ? __builtin_amdgcn_update_dpp(de, xe, 0x138, 0xf, 0xf, false) : xe + 1;
                const int kai = k & 3, kti = k & 1;
                const int ex = ue + kai, ey = xe + kai, ed = de;
                const int er = max(max(ex, ey), ed);
                const float u = __builtin_amdgcn_ldexpf(ua, ex - er);
                const float l = __builtin_amdgcn_ldexpf(x, ey - er);
                const float dd = __builtin_amdgcn_ldexpf(d, ed - er);
                const float ssum = __builtin_fmaf(ca, u + l, dd);
                if (MODE & 2) {
                    const float tq = ca * __builtin_amdgcn_rcpf(ssum);
                    acc += tq * u + tq * l;
                }
                const float an = ct * ssum;
                d = ua;
                de = ue;
                x = __builtin_amdgcn_frexp_mantf(an);
                xe = er + kti + __builtin_amdgcn_frexp_expf(an);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + d + acc + (float)mx + (float)mn + (float)xe;
}

template <int MODE>
float run(int wg, int waves, int steps, float *out, float *in)
{
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    chain<MODE><<<wg, waves * 64>>>(out, in, steps);
    hipEventRecord(a);
    chain<MODE><<<wg, waves * 64>>>(out, in, steps);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms * 1e6f / steps;  // ns per step
}

int main(int argc, char **argv)
{
    float *out, *in;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&in, 64 * 4);
    float h[64];
    for (int i = 0; i < 64; ++i) h[i] = 0.01f * i;
    h[0] = 0.f;
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    const int steps = 1 << 16;
    for (int waves : {1, 4, 8}) {
        printf("waves/WG=%d (one WG per CU): ns/step  (cycles @2.4GHz)\n", waves);
#define R(M, name) { float t = run<M>(256, waves, steps, out, in); printf("  %-44s %7.1f  (%5.0f)\n", name, t, t * 2.4f); }
        R(0, "wf chain only (mul add fma mul, no dpp)");
        R(4, "wf chain + dpp");
        R(4 | 1, "wf chain + dpp + 2 exp");
        R(4 | 1 | 2, "wf chain + dpp + 2 exp + rcp/q");
        R(4 | 1 | 2 | 8, "wf full (with range tracking)");
        R(16, "slow chain only (no dpp)");
        R(16 | 4, "slow chain + 2 dpp");
        R(16 | 4 | 1 | 2, "slow full");
    }
    return 0;
}
