// probe: raw buffer dwordx4 loads/stores -- dword-aligned addresses and partial out-of-range behaviour
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, unsigned bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), (short)0, (int)bytes, 0x00020000);
}
__global__ void probe(const unsigned *in, unsigned *out, unsigned *st)
{
    auto rs = make_rsrc(in, 6 * 4);  // 6 dwords visible
    u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, 4, 0, 0);    // dwords 1..4 (unaligned to 16)
    u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, 16, 0, 0);   // dwords 4,5 in range, 6,7 out
    u32x4 c = __builtin_amdgcn_raw_buffer_load_b128(rs, 0, 12, 0);   // via scalar offset: dwords 3..6
    u32x4 d = __builtin_amdgcn_raw_buffer_load_b128(rs, 0xfffffff0u, 0, 0);  // "negative" offset
    if (threadIdx.x == 0) {
        for (int i = 0; i < 4; ++i) out[i] = a[i], out[4 + i] = b[i], out[8 + i] = c[i], out[12 + i] = d[i];
    }
    auto ws = make_rsrc(st, 6 * 4);
    u32x4 v = {101, 102, 103, 104};
    if (threadIdx.x == 0) __builtin_amdgcn_raw_buffer_store_b128(v, ws, 16, 0, 0);  // dwords 4,5 in, 6,7 out
    if (threadIdx.x == 1) __builtin_amdgcn_raw_buffer_store_b128(v, ws, 4, 0, 0);   // dwords 1..4; lane 0 overwrites 4
}
int main()
{
    unsigned h[16], *din, *dout, *dst, ho[16], hs[8];
    for (int i = 0; i < 16; ++i) h[i] = 10 + i;
    hipMalloc(&din, 64); hipMalloc(&dout, 64); hipMalloc(&dst, 32);
    hipMemcpy(din, h, 64, hipMemcpyHostToDevice);
    hipMemset(dst, 0, 32);
    probe<<<1, 64>>>(din, dout, dst);
    hipMemcpy(ho, dout, 64, hipMemcpyDeviceToHost);
    hipMemcpy(hs, dst, 32, hipMemcpyDeviceToHost);
    printf("unaligned  : %u %u %u %u (want 11 12 13 14)\n", ho[0], ho[1], ho[2], ho[3]);
    printf("partial oob: %u %u %u %u (per-dword: 14 15 0 0)\n", ho[4], ho[5], ho[6], ho[7]);
    printf("soffset    : %u %u %u %u (per-dword: 13 14 15 0)\n", ho[8], ho[9], ho[10], ho[11]);
    printf("negative   : %u %u %u %u (want 0 0 0 0)\n", ho[12], ho[13], ho[14], ho[15]);
    printf("store      : %u %u %u %u %u %u %u %u (per-dword: 0 101 102 103 10x 102 0 0)\n", hs[0], hs[1], hs[2], hs[3], hs[4], hs[5], hs[6], hs[7]);
    return 0;
}
