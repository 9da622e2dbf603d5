// probe: do ds_read_b128 / ds_write_b128 work at addresses that are only 4-byte aligned, and what do they cost?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(float *out, long long *cyc, int shift_mode)
{
    __shared__ __attribute__((aligned(16))) float buf[64 * 72 + 16];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 72 + 16; i += 64) buf[i] = -1.f;
    __syncthreads();
    // lane l writes 4 floats at l*72 + 8 + s, s = (l % 4) in shift mode, else 0 (pitch 72: conflict-free-ish)
    const int s = shift_mode ? (lane & 3) : 0;
    unsigned addr = (unsigned)(uintptr_t)(buf + lane * 72 + 8 + s);
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 w = {lane * 10.f, lane * 10.f + 1, lane * 10.f + 2, lane * 10.f + 3};
    long long t0 = clock64();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
        asm volatile("ds_write_b128 %0, %1 offset:64\n\ts_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(w) : "memory");
    }
    long long t1 = clock64();
    asm volatile("ds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : : "v"(addr), "v"(w) : "memory");
    __syncthreads();
    float4 acc = make_float4(0, 0, 0, 0);
    long long t2 = clock64();
#pragma unroll 1
    for (int it = 0; it < 256; ++it) {
        f32x4 v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    long long t3 = clock64();
    for (int j = 0; j < 8; ++j) out[lane * 8 + j] = buf[lane * 72 + 8 + j];
    if (lane == 0) { cyc[2 * shift_mode] = t1 - t0; cyc[2 * shift_mode + 1] = t3 - t2; }
    out[600 + lane] = acc.x + acc.y + acc.z + acc.w;
}
int main()
{
    float *out; long long *cyc; float h[512]; long long hc[4];
    hipMalloc(&out, 8192); hipMalloc(&cyc, 32);
    for (int mode = 0; mode < 2; ++mode) {
        probe<<<1, 64>>>(out, cyc, mode);
        hipDeviceSynchronize();
        hipMemcpy(h, out, 2048, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 8; ++j) {
                const int s = mode ? (l & 3) : 0;
                const float want = (j >= s && j < s + 4) ? l * 10.f + (j - s) : -1.f;
                if (h[l * 8 + j] != want) ++bad;
            }
        hipMemcpy(hc, cyc, 32, hipMemcpyDeviceToHost);
        printf("mode %d (%s): wrong values after write %d; ticks per write %.1f, per read %.1f\n", mode, mode ? "4-byte aligned" : "16-byte aligned", bad,
               hc[2 * mode] / 256.0, hc[2 * mode + 1] / 256.0);
    }
    return 0;
}
