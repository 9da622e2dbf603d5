// Issue cost of the instructions the sweeps are made of, per wave-instruction in shader cycles (s_memtime), for
//   * an independent stream (8 destination registers in rotation) and a dependent chain (one register),
//   * 1, 2 (and 4) waves per SIMD (workgroups of 4 / 8 / 16 waves, one workgroup per CU).
// usage: isa_cost            -- prints one table; hipcc --offload-arch=gfx950 -O3 isa_cost.hip -o isa_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// each TEST: asm text for "independent" (uses %0..%7 as dst, %8 %9 as extra sources) and "dependent" (only %0)
#define DEF_TEST(ID, NAME, IND, DEP)                                                                       \
    template <> struct Test<ID> {                                                                          \
        static constexpr const char *name = NAME;                                                          \
        __device__ static void ind(float (&v)[8], float a, float b) {                                      \
            asm volatile(IND IND IND IND                                                                   \
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) \
                         : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "vcc", "scc");             \
        }                                                                                                  \
        __device__ static void dep(float (&v)[8], float a, float b) {                                      \
            asm volatile(DEP DEP DEP DEP                                                                   \
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) \
                         : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "vcc", "scc");             \
        }                                                                                                  \
    };
template <int ID> struct Test;

#define I3(op) op " %0, %0, %8, %0\n" op " %1, %1, %8, %1\n" op " %2, %2, %8, %2\n" op " %3, %3, %8, %3\n" op " %4, %4, %8, %4\n" op " %5, %5, %8, %5\n" op " %6, %6, %8, %6\n" op " %7, %7, %8, %7\n"
#define D3(op) op " %0, %0, %8, %0\n" op " %0, %0, %8, %0\n" op " %0, %0, %8, %0\n" op " %0, %0, %8, %0\n" op " %0, %0, %8, %0\n" op " %0, %0, %8, %0\n" op " %0, %0, %8, %0\n" op " %0, %0, %8, %0\n"
#define I2(op) op " %0, %0, %8\n" op " %1, %1, %8\n" op " %2, %2, %8\n" op " %3, %3, %8\n" op " %4, %4, %8\n" op " %5, %5, %8\n" op " %6, %6, %8\n" op " %7, %7, %8\n"
#define D2(op) op " %0, %0, %8\n" op " %0, %0, %8\n" op " %0, %0, %8\n" op " %0, %0, %8\n" op " %0, %0, %8\n" op " %0, %0, %8\n" op " %0, %0, %8\n" op " %0, %0, %8\n"
#define I1(op) op " %0, %0\n" op " %1, %1\n" op " %2, %2\n" op " %3, %3\n" op " %4, %4\n" op " %5, %5\n" op " %6, %6\n" op " %7, %7\n"
#define D1(op) op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n" op " %0, %0\n"
#define I1S(op, suf) op " %0, %0 " suf "\n" op " %1, %1 " suf "\n" op " %2, %2 " suf "\n" op " %3, %3 " suf "\n" op " %4, %4 " suf "\n" op " %5, %5 " suf "\n" op " %6, %6 " suf "\n" op " %7, %7 " suf "\n"
#define D1S(op, suf) op " %0, %0 " suf "\n" op " %0, %0 " suf "\n" op " %0, %0 " suf "\n" op " %0, %0 " suf "\n" op " %0, %0 " suf "\n" op " %0, %0 " suf "\n" op " %0, %0 " suf "\n" op " %0, %0 " suf "\n"
// cross: dst i <- src (i+1): the DPP source is another register (as in the sweep: carry of the neighbour lane)
#define X1S(op, suf) op " %0, %1 " suf "\n" op " %1, %2 " suf "\n" op " %2, %3 " suf "\n" op " %3, %4 " suf "\n" op " %4, %5 " suf "\n" op " %5, %6 " suf "\n" op " %6, %7 " suf "\n" op " %7, %0 " suf "\n"

DEF_TEST(0, "v_fma_f32", I3("v_fma_f32"), D3("v_fma_f32"))
DEF_TEST(1, "v_mul_f32", I2("v_mul_f32"), D2("v_mul_f32"))
DEF_TEST(2, "v_add_f32", I2("v_add_f32"), D2("v_add_f32"))
DEF_TEST(3, "v_exp_f32", I1("v_exp_f32"), D1("v_exp_f32"))
DEF_TEST(4, "v_rcp_f32", I1("v_rcp_f32"), D1("v_rcp_f32"))
DEF_TEST(5, "v_log_f32", I1("v_log_f32"), D1("v_log_f32"))
DEF_TEST(6, "v_ldexp_f32", I2("v_ldexp_f32"), D2("v_ldexp_f32"))
DEF_TEST(7, "v_frexp_mant_f32", I1("v_frexp_mant_f32"), D1("v_frexp_mant_f32"))
DEF_TEST(8, "v_mov_b32_dpp wave_shr:1", X1S("v_mov_b32_dpp", "wave_shr:1 row_mask:0xf bank_mask:0xf"), D1S("v_mov_b32_dpp", "wave_shr:1 row_mask:0xf bank_mask:0xf"))
DEF_TEST(9, "v_mov_b32_dpp row_shr:1", X1S("v_mov_b32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf"), D1S("v_mov_b32_dpp", "row_shr:1 row_mask:0xf bank_mask:0xf"))
DEF_TEST(10, "v_max_u32", I2("v_max_u32"), D2("v_max_u32"))
DEF_TEST(11, "v_max3_u32", I3("v_max3_u32"), D3("v_max3_u32"))
DEF_TEST(12, "v_perm_b32", I3("v_perm_b32"), D3("v_perm_b32"))
DEF_TEST(13, "v_cndmask_b32 (vcc)", I2("v_cndmask_b32_e32"), D2("v_cndmask_b32_e32"))
DEF_TEST(14, "v_mov_b32", I1("v_mov_b32"), D1("v_mov_b32"))
DEF_TEST(15, "v_add_u32", I2("v_add_u32"), D2("v_add_u32"))
DEF_TEST(16, "v_mul_f32_dpp wave_shr:1 (fused)", "v_mul_f32_dpp %0, %1, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mul_f32_dpp %1, %2, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mul_f32_dpp %2, %3, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mul_f32_dpp %3, %4, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mul_f32_dpp %4, %5, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mul_f32_dpp %5, %6, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mul_f32_dpp %6, %7, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mul_f32_dpp %7, %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n" , "v_mul_f32_dpp %0, %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mul_f32_dpp %0, %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mul_f32_dpp %0, %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mul_f32_dpp %0, %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mul_f32_dpp %0, %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mul_f32_dpp %0, %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mul_f32_dpp %0, %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n" "v_mul_f32_dpp %0, %0, %8 wave_shr:1 row_mask:0xf bank_mask:0xf\n" )
DEF_TEST(17, "v_fmac_f32", I2("v_fmac_f32"), D2("v_fmac_f32"))
DEF_TEST(18, "v_cmp_lt_f32 (-> vcc)", "v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n", "v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %0, %8\n")
DEF_TEST(19, "s_nop 0", "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n", "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n")
DEF_TEST(20, "v_readfirstlane_b32 (-> s)", "v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s20, %2\n v_readfirstlane_b32 s21, %3\n v_readfirstlane_b32 s20, %4\n v_readfirstlane_b32 s21, %5\n v_readfirstlane_b32 s20, %6\n v_readfirstlane_b32 s21, %7\n", "v_readfirstlane_b32 s20, %0\n v_mov_b32 %0, s20\n v_readfirstlane_b32 s20, %0\n v_mov_b32 %0, s20\n v_readfirstlane_b32 s20, %0\n v_mov_b32 %0, s20\n v_readfirstlane_b32 s20, %0\n v_mov_b32 %0, s20\n")

DEF_TEST(21, "v_cndmask_b32_e64 (sgpr-pair mask)", "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n" "v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n" "v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n" "v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n" "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n" "v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n" "v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n" "v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n" , "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n" "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n" "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n" "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n" "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n" "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n" "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n" "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n" )
DEF_TEST(22, "v_cmp_lt_f32 + v_cndmask (per pair)", "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32_e32 %0, %0, %8\n" "v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32_e32 %1, %1, %8\n" "v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32_e32 %2, %2, %8\n" "v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32_e32 %3, %3, %8\n" "v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32_e32 %4, %4, %8\n" "v_cmp_lt_f32 vcc, %5, %8\n v_cndmask_b32_e32 %5, %5, %8\n" "v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32_e32 %6, %6, %8\n" "v_cmp_lt_f32 vcc, %7, %8\n v_cndmask_b32_e32 %7, %7, %8\n" , "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32_e32 %0, %0, %8\n" "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32_e32 %0, %0, %8\n" "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32_e32 %0, %0, %8\n" "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32_e32 %0, %0, %8\n" "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32_e32 %0, %0, %8\n" "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32_e32 %0, %0, %8\n" "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32_e32 %0, %0, %8\n" "v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32_e32 %0, %0, %8\n" )
DEF_TEST(23, "v_bfi_b32", "v_bfi_b32 %0, %8, %0, %9\n" "v_bfi_b32 %1, %8, %1, %9\n" "v_bfi_b32 %2, %8, %2, %9\n" "v_bfi_b32 %3, %8, %3, %9\n" "v_bfi_b32 %4, %8, %4, %9\n" "v_bfi_b32 %5, %8, %5, %9\n" "v_bfi_b32 %6, %8, %6, %9\n" "v_bfi_b32 %7, %8, %7, %9\n" , "v_bfi_b32 %0, %8, %0, %9\n" "v_bfi_b32 %0, %8, %0, %9\n" "v_bfi_b32 %0, %8, %0, %9\n" "v_bfi_b32 %0, %8, %0, %9\n" "v_bfi_b32 %0, %8, %0, %9\n" "v_bfi_b32 %0, %8, %0, %9\n" "v_bfi_b32 %0, %8, %0, %9\n" "v_bfi_b32 %0, %8, %0, %9\n" )
DEF_TEST(24, "v_med3_f32", "v_med3_f32 %0, %0, %8, %9\n" "v_med3_f32 %1, %1, %8, %9\n" "v_med3_f32 %2, %2, %8, %9\n" "v_med3_f32 %3, %3, %8, %9\n" "v_med3_f32 %4, %4, %8, %9\n" "v_med3_f32 %5, %5, %8, %9\n" "v_med3_f32 %6, %6, %8, %9\n" "v_med3_f32 %7, %7, %8, %9\n" , "v_med3_f32 %0, %0, %8, %9\n" "v_med3_f32 %0, %0, %8, %9\n" "v_med3_f32 %0, %0, %8, %9\n" "v_med3_f32 %0, %0, %8, %9\n" "v_med3_f32 %0, %0, %8, %9\n" "v_med3_f32 %0, %0, %8, %9\n" "v_med3_f32 %0, %0, %8, %9\n" "v_med3_f32 %0, %0, %8, %9\n" )
DEF_TEST(25, "s_add_u32", "s_add_u32 s20, s20, 1\n" "s_add_u32 s21, s21, 1\n" "s_add_u32 s22, s22, 1\n" "s_add_u32 s23, s23, 1\n" "s_add_u32 s24, s24, 1\n" "s_add_u32 s25, s25, 1\n" "s_add_u32 s26, s26, 1\n" "s_add_u32 s27, s27, 1\n" , "s_add_u32 s20, s20, 1\n" "s_add_u32 s20, s20, 1\n" "s_add_u32 s20, s20, 1\n" "s_add_u32 s20, s20, 1\n" "s_add_u32 s20, s20, 1\n" "s_add_u32 s20, s20, 1\n" "s_add_u32 s20, s20, 1\n" "s_add_u32 s20, s20, 1\n" )
DEF_TEST(26, "s_and_b64", "s_and_b64 s[20:21], s[20:21], s[22:23]\n" "s_and_b64 s[20:21], s[20:21], s[22:23]\n" "s_and_b64 s[20:21], s[20:21], s[22:23]\n" "s_and_b64 s[20:21], s[20:21], s[22:23]\n" "s_and_b64 s[20:21], s[20:21], s[22:23]\n" "s_and_b64 s[20:21], s[20:21], s[22:23]\n" "s_and_b64 s[20:21], s[20:21], s[22:23]\n" "s_and_b64 s[20:21], s[20:21], s[22:23]\n" , "s_and_b64 s[20:21], s[20:21], s[22:23]\n" "s_and_b64 s[20:21], s[20:21], s[22:23]\n" "s_and_b64 s[20:21], s[20:21], s[22:23]\n" "s_and_b64 s[20:21], s[20:21], s[22:23]\n" "s_and_b64 s[20:21], s[20:21], s[22:23]\n" "s_and_b64 s[20:21], s[20:21], s[22:23]\n" "s_and_b64 s[20:21], s[20:21], s[22:23]\n" "s_and_b64 s[20:21], s[20:21], s[22:23]\n" )
DEF_TEST(27, "v_cmp_lt_f32_e64 (-> sgpr pair)", "v_cmp_lt_f32_e64 s[20:21], %0, %8\n" "v_cmp_lt_f32_e64 s[20:21], %1, %8\n" "v_cmp_lt_f32_e64 s[20:21], %2, %8\n" "v_cmp_lt_f32_e64 s[20:21], %3, %8\n" "v_cmp_lt_f32_e64 s[20:21], %4, %8\n" "v_cmp_lt_f32_e64 s[20:21], %5, %8\n" "v_cmp_lt_f32_e64 s[20:21], %6, %8\n" "v_cmp_lt_f32_e64 s[20:21], %7, %8\n" , "v_cmp_lt_f32_e64 s[20:21], %0, %8\n" "v_cmp_lt_f32_e64 s[20:21], %0, %8\n" "v_cmp_lt_f32_e64 s[20:21], %0, %8\n" "v_cmp_lt_f32_e64 s[20:21], %0, %8\n" "v_cmp_lt_f32_e64 s[20:21], %0, %8\n" "v_cmp_lt_f32_e64 s[20:21], %0, %8\n" "v_cmp_lt_f32_e64 s[20:21], %0, %8\n" "v_cmp_lt_f32_e64 s[20:21], %0, %8\n" )
DEF_TEST(28, "v_mov_b32 x2 alternating with s_add (mix)", "v_mov_b32 %0, %0\n s_add_u32 s20, s20, 1\n" "v_mov_b32 %1, %1\n s_add_u32 s20, s20, 1\n" "v_mov_b32 %2, %2\n s_add_u32 s20, s20, 1\n" "v_mov_b32 %3, %3\n s_add_u32 s20, s20, 1\n" "v_mov_b32 %4, %4\n s_add_u32 s20, s20, 1\n" "v_mov_b32 %5, %5\n s_add_u32 s20, s20, 1\n" "v_mov_b32 %6, %6\n s_add_u32 s20, s20, 1\n" "v_mov_b32 %7, %7\n s_add_u32 s20, s20, 1\n" , "v_mov_b32 %0, %0\n s_add_u32 s20, s20, 1\n" "v_mov_b32 %0, %0\n s_add_u32 s20, s20, 1\n" "v_mov_b32 %0, %0\n s_add_u32 s20, s20, 1\n" "v_mov_b32 %0, %0\n s_add_u32 s20, s20, 1\n" "v_mov_b32 %0, %0\n s_add_u32 s20, s20, 1\n" "v_mov_b32 %0, %0\n s_add_u32 s20, s20, 1\n" "v_mov_b32 %0, %0\n s_add_u32 s20, s20, 1\n" "v_mov_b32 %0, %0\n s_add_u32 s20, s20, 1\n" )

template <int ID, bool DEP>
__global__ void __launch_bounds__(1024) bench(unsigned long long *out, float a, float b, int iters)
{
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = 1.0f + 0.001f * (threadIdx.x + i);
    unsigned long long t0 = __builtin_readcyclecounter();
    t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (DEP) Test<ID>::dep(v, a, b);
        else Test<ID>::ind(v, a, b);
    }
    unsigned long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678f) out[1] = 1;
    if ((threadIdx.x & 63) == 0) out[2 + blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

// v_pk_fma_f32 / f64 ops need register pairs: separate kernels
template <int KIND, bool DEP>
__global__ void __launch_bounds__(1024) bench2(unsigned long long *out, double a, int iters)
{
    double v[8];
    for (int i = 0; i < 8; ++i) v[i] = 1.0 + 0.001 * (threadIdx.x + i);
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#define PK(op, A, B, C) op " %" #A ", %" #B ", %8, %" #C "\n"
        if (KIND == 0) {  // v_pk_fma_f32 (two floats per lane and instruction)
            if (DEP) asm volatile(PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0)
                                  PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0)
                                  PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0)
                                  PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",0,0,0)
                                  : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(a));
            else asm volatile(PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",1,1,1) PK("v_pk_fma_f32",2,2,2) PK("v_pk_fma_f32",3,3,3) PK("v_pk_fma_f32",4,4,4) PK("v_pk_fma_f32",5,5,5) PK("v_pk_fma_f32",6,6,6) PK("v_pk_fma_f32",7,7,7)
                              PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",1,1,1) PK("v_pk_fma_f32",2,2,2) PK("v_pk_fma_f32",3,3,3) PK("v_pk_fma_f32",4,4,4) PK("v_pk_fma_f32",5,5,5) PK("v_pk_fma_f32",6,6,6) PK("v_pk_fma_f32",7,7,7)
                              PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",1,1,1) PK("v_pk_fma_f32",2,2,2) PK("v_pk_fma_f32",3,3,3) PK("v_pk_fma_f32",4,4,4) PK("v_pk_fma_f32",5,5,5) PK("v_pk_fma_f32",6,6,6) PK("v_pk_fma_f32",7,7,7)
                              PK("v_pk_fma_f32",0,0,0) PK("v_pk_fma_f32",1,1,1) PK("v_pk_fma_f32",2,2,2) PK("v_pk_fma_f32",3,3,3) PK("v_pk_fma_f32",4,4,4) PK("v_pk_fma_f32",5,5,5) PK("v_pk_fma_f32",6,6,6) PK("v_pk_fma_f32",7,7,7)
                              : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(a));
        } else {  // v_fma_f64
            if (DEP) asm volatile(PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0)
                                  PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0)
                                  PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0)
                                  PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0) PK("v_fma_f64",0,0,0)
                                  : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(a));
            else asm volatile(PK("v_fma_f64",0,0,0) PK("v_fma_f64",1,1,1) PK("v_fma_f64",2,2,2) PK("v_fma_f64",3,3,3) PK("v_fma_f64",4,4,4) PK("v_fma_f64",5,5,5) PK("v_fma_f64",6,6,6) PK("v_fma_f64",7,7,7)
                              PK("v_fma_f64",0,0,0) PK("v_fma_f64",1,1,1) PK("v_fma_f64",2,2,2) PK("v_fma_f64",3,3,3) PK("v_fma_f64",4,4,4) PK("v_fma_f64",5,5,5) PK("v_fma_f64",6,6,6) PK("v_fma_f64",7,7,7)
                              PK("v_fma_f64",0,0,0) PK("v_fma_f64",1,1,1) PK("v_fma_f64",2,2,2) PK("v_fma_f64",3,3,3) PK("v_fma_f64",4,4,4) PK("v_fma_f64",5,5,5) PK("v_fma_f64",6,6,6) PK("v_fma_f64",7,7,7)
                              PK("v_fma_f64",0,0,0) PK("v_fma_f64",1,1,1) PK("v_fma_f64",2,2,2) PK("v_fma_f64",3,3,3) PK("v_fma_f64",4,4,4) PK("v_fma_f64",5,5,5) PK("v_fma_f64",6,6,6) PK("v_fma_f64",7,7,7)
                              : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(a));
        }
    }
    unsigned long long t1 = clock64();
    double s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    if (s == 12345.678) out[1] = 1;
    if ((threadIdx.x & 63) == 0) out[2 + blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

// LDS: ds_read_b128 / ds_read_b32 bursts of 8 + one wait, ds_write_b32
template <int KIND>
__global__ void __launch_bounds__(1024) bench_lds(unsigned long long *out, int iters)
{
    __shared__ float4 buf[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) buf[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)buf + (threadIdx.x & 1023) * 16;
    float4 r[8];
    float acc = 0;
    unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
            asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16384\n ds_read_b128 %2, %8 offset:32768\n ds_read_b128 %3, %8 offset:49152\n"
                         "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:16448\n ds_read_b128 %6, %8 offset:32832\n ds_read_b128 %7, %8 offset:49216\n s_waitcnt lgkmcnt(0)\n"
                         : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]) : "v"(base));
            acc += r[0].x + r[7].w;
        } else if (KIND == 1) {
            float q[8];
            asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:4096\n ds_read_b32 %2, %8 offset:8192\n ds_read_b32 %3, %8 offset:12288\n"
                         "ds_read_b32 %4, %8 offset:64\n ds_read_b32 %5, %8 offset:4160\n ds_read_b32 %6, %8 offset:8256\n ds_read_b32 %7, %8 offset:12352\n s_waitcnt lgkmcnt(0)\n"
                         : "=v"(q[0]), "=v"(q[1]), "=v"(q[2]), "=v"(q[3]), "=v"(q[4]), "=v"(q[5]), "=v"(q[6]), "=v"(q[7]) : "v"(base));
            acc += q[0] + q[7];
        } else {
            asm volatile("ds_write_b32 %0, %1\n ds_write_b32 %0, %1 offset:4096\n ds_write_b32 %0, %1 offset:8192\n ds_write_b32 %0, %1 offset:12288\n"
                         "ds_write_b32 %0, %1 offset:64\n ds_write_b32 %0, %1 offset:4160\n ds_write_b32 %0, %1 offset:8256\n ds_write_b32 %0, %1 offset:12352\n s_waitcnt lgkmcnt(0)\n"
                         : : "v"(base), "v"(acc) : "memory");
        }
    }
    unsigned long long t1 = clock64();
    if (acc == 12345.678f) out[1] = 1;
    if ((threadIdx.x & 63) == 0) out[2 + blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

static unsigned long long *d_out;
static std::vector<unsigned long long> h_out(2 + 256 * 16);

template <class F>
double run(F launch, int waves, int per_iter, int iters)
{
    hipMemset(d_out, 0, h_out.size() * 8);
    launch(waves, iters);
    hipDeviceSynchronize();
    hipMemcpy(h_out.data(), d_out, h_out.size() * 8, hipMemcpyDeviceToHost);
    double sum = 0;
    int n = 0;
    for (int b = 0; b < 256; ++b)
        for (int w = 0; w < waves; ++w) sum += (double)h_out[2 + b * 16 + w], ++n;
    return sum / n / ((double)iters * per_iter);
}

template <int ID>
void row()
{
    const int iters = 2000;
    printf("%-40s", Test<ID>::name);
    for (int dep = 0; dep < 2; ++dep)
        for (int waves : {4, 8, 16}) {
            double c = dep ? run([](int w, int it) { bench<ID, true><<<256, w * 64>>>(d_out, 1.0001f, 0.5f, it); }, waves, 32, iters)
                           : run([](int w, int it) { bench<ID, false><<<256, w * 64>>>(d_out, 1.0001f, 0.5f, it); }, waves, 32, iters);
            printf(" %7.2f", c);
        }
    printf("\n");
}

template <int... IDS>
void rows() { (row<IDS>(), ...); }

int main()
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    hipMalloc(&d_out, h_out.size() * 8);
    // clock64 = s_memtime: report the ratio to wall clock as well
    {
        hipEvent_t a, b;
        hipEventCreate(&a), hipEventCreate(&b);
        bench<0, true><<<256, 256>>>(d_out, 1.f, .5f, 100);
        hipEventRecord(a);
        bench<0, true><<<256, 256>>>(d_out, 1.f, .5f, 20000);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        hipMemcpy(h_out.data(), d_out, h_out.size() * 8, hipMemcpyDeviceToHost);
        printf("clock64 ticks per us: %.1f (kernel %.3f ms, %llu ticks)\n", h_out[2] / (ms * 1e3), ms, h_out[2]);
    }
    printf("cycles (clock64 ticks) per wave-instruction; waves per CU = 4 / 8 / 16 (1 / 2 / 4 per SIMD)\n");
    printf("%-40s %7s %7s %7s %7s %7s %7s\n", "instruction", "ind:4", "ind:8", "ind:16", "dep:4", "dep:8", "dep:16");
    rows<0, 1, 2, 17, 3, 4, 5, 6, 7, 8, 9, 16, 10, 11, 12, 13, 21, 22, 27, 23, 24, 14, 15, 18, 19, 20, 25, 26, 28>();
    const int iters = 2000;
    for (int kind = 0; kind < 2; ++kind) {
        printf("%-40s", kind == 0 ? "v_pk_fma_f32 (2 fp32 fma per lane)" : "v_fma_f64");
        for (int dep = 0; dep < 2; ++dep)
            for (int waves : {4, 8, 16}) {
                double c;
                if (kind == 0) c = dep ? run([](int w, int it) { bench2<0, true><<<256, w * 64>>>(d_out, 1.0001, it); }, waves, 32, iters)
                                       : run([](int w, int it) { bench2<0, false><<<256, w * 64>>>(d_out, 1.0001, it); }, waves, 32, iters);
                else c = dep ? run([](int w, int it) { bench2<1, true><<<256, w * 64>>>(d_out, 1.0001, it); }, waves, 32, iters)
                             : run([](int w, int it) { bench2<1, false><<<256, w * 64>>>(d_out, 1.0001, it); }, waves, 32, iters);
                printf(" %7.2f", c);
            }
        printf("\n");
    }
    const char *ln[3] = {"ds_read_b128 x8 + wait (per instr)", "ds_read_b32 x8 + wait (per instr)", "ds_write_b32 x8 + wait (per instr)"};
    for (int kind = 0; kind < 3; ++kind) {
        printf("%-40s", ln[kind]);
        for (int waves : {4, 8, 16}) {
            double c = kind == 0 ? run([](int w, int it) { bench_lds<0><<<256, w * 64>>>(d_out, it); }, waves, 8, iters)
                     : kind == 1 ? run([](int w, int it) { bench_lds<1><<<256, w * 64>>>(d_out, it); }, waves, 8, iters)
                                 : run([](int w, int it) { bench_lds<2><<<256, w * 64>>>(d_out, it); }, waves, 8, iters);
            printf(" %7.2f", c);
        }
        printf("\n");
    }
    return 0;
}
