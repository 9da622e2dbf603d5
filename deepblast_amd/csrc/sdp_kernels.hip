// sdp_kernels.hip -- wavefront-skewed soft-DP alignment kernels for gfx950 (MI355X).
//
// Replaces the four Numba-CUDA kernels of the reference (one thread per pair, serial
// O(N*M) loop: deepblast/nw_cuda.py:46-165, sw_cuda.py:46-165) with an anti-diagonal
// sweep designed for CDNA4.  Results follow the CPU reference deepblast/nw.py (A indexed
// [i-1,j-1], nw.py:56-58 -- NOT the GPU reference's A[last, j-1], nw_cuda.py:61-63) to
// <= 1e-4; see "carries" below for the arithmetic each pass uses.
//
// Mapping:
//   * one workgroup per pair, W <= 8 wavefronts (throughput builds: K = 32, <= 4 waves; latency builds:
//     K = 16, <= 8 waves; the host picks per launch, sdp_api.hip::plan);
//   * the N rows are cut into strips of 64; wave w owns strips w, w+W, ...;
//   * inside a strip lane l owns row i0+l and at step t sits on column t-l, so the
//     64 lanes of a wave always lie on one anti-diagonal.  The two predecessors from
//     row i-1 arrive from lane l-1 through one DPP full-wave shift of the carry (no
//     LDS, no barrier); the row-i predecessor is the lane's own register;
//   * strip-to-strip hand-off (bottom row of strip s -> lane 0 of strip s+1) goes
//     through an 8-byte-slot row buffer in LDS, published K columns at a time (forward
//     sweep: 16 at a time, see fwd_blocks) with a monotonic progress word per wave (no
//     s_barrier anywhere in the sweep);
//   * row-major tensors (theta, A, Ztheta, ZA, E, Ed) cross the skew through LDS:
//     inputs as K-column blocks loaded four columns per lane into a rotated per-row
//     ring that the lanes read back with aligned 16-byte reads, prefetched through
//     registers one chunk ahead; outputs are written to LDS by step and leave as
//     blocks; output blocks (and the input blocks of the throughput forward build) are
//     aligned to 128-byte lines of MEMORY whatever the row pitch is (per-row offsets);
//   * the forward sweep computes in blocks of 16 steps inside a K-step chunk: inputs,
//     boundary values and published values live in 16-entry arrays (fwd_blocks);
//   * the saved state (reference: Q, (B,N+2,M+2,3) fp32) is private to this library,
//     so it is stored ALREADY SKEWED, [pair][strip][t][lane]: packed to 5 bytes per
//     cell for the backward sweep (two 20-bit fields; 16 steps of a lane = five dwordx4), or float2 for the adjoint
//     sweeps (qm = 1 - qx - qy is never stored).  Forward writes and backward reads
//     are then fully coalesced wave accesses with no transposition;
//   * the forward recurrence runs in a scaled exp domain; normally in its windowed
//     form (one exponent per lane and 16-step block, see steps_wf), with the
//     per-step-normalised form as the verified fallback;
//   * the reverse passes run the same (t, lane) -> cell schedule backwards in "push"
//     form: each cell scales its E by its own three weights and hands the products to
//     its predecessors, so every cell's weights are read exactly once, by its owner.
//
// No MFMA: this is a scalar recurrence, bounded by HBM bytes and by the length of the
// dependency chain (N+M-1 steps), not by matrix throughput.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "sdp_kernels.h"

namespace sdp {

// ----------------------------------------------------------------------------------
// cross-lane moves (DPP full-wave shifts/rotates; gfx9 family encodings)
// ----------------------------------------------------------------------------------
constexpr int DPP_WAVE_SHL1 = 0x130;  // lane i <- lane i+1 ; lane 63 keeps `old`
constexpr int DPP_WAVE_ROL1 = 0x134;  // lane i <- lane (i+1)%64
constexpr int DPP_WAVE_SHR1 = 0x138;  // lane i <- lane i-1 ; lane 0 keeps `old`
constexpr int DPP_WAVE_ROR1 = 0x13C;  // lane i <- lane (i-1)%64

template <int CTRL>
__device__ __forceinline__ int dpp_i32(int old, int src)
{
    return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xf, 0xf, false);
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double old, double src)
{
    const int lo = dpp_i32<CTRL>(__double2loint(old), __double2loint(src));
    const int hi = dpp_i32<CTRL>(__double2hiint(old), __double2hiint(src));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ float fast_exp(float x)  // e^x via v_exp_f32 (2^x)
{
    return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
}
__device__ __forceinline__ float fast_log(float x)  // ln x via v_log_f32 (log2 x)
{
    return __builtin_amdgcn_logf(x) * 0.69314718055994530942f;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes)
{
    // raw buffer (stride 0), 32-bit data format; out-of-range loads return 0, stores are dropped
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

constexpr unsigned OOB = 0x80000000u;  // voffset that is out of range for every buffer we build

// Ablation switches for timing experiments (tools/gpu_tune.py builds variants with -DSDP_ABL=mask);
// results are wrong when any bit is set.  bit0: no global stores, bit1: no global loads, bit2: no strip
// hand-off waits, bit3: keep every load/store but replace the recurrence by a copy.
#ifndef SDP_ABL
#define SDP_ABL 0
#endif
#ifdef SDP_EXPERIMENTS
#define SDP_EXP_BUILD 1
#else
#define SDP_EXP_BUILD 0  // default library: Params::dbg is ignored, no wrong-results switch is reachable
#endif
#ifndef SDP_TB_WINDOW
#define SDP_TB_WINDOW 32  // traceback: edge of the LDS window of E (32 or 64 cells)
#endif
// Everything below used to be a compile-time switch of its own (31 of them by round 5).  Each was measured, one setting won, and
// the losing code paths were deleted in round 6 (their measurements: DESIGN_HISTORY.md, "Switches retired in round 6"; the code:
// git history).  What is left are the constants the winning settings fold to.
//   * fp32 backward sweep: a chunk whose carries, boundary values and cotangent are all +0 produces +0 everywhere: its steps are
//     skipped and its state rows not loaded (bit-identical; the control is the run-time flag SDP_NO_ZERO_SKIP);
//   * adjoint backward sweep: chunks over which E, the carries and the boundary values are all zero are not run (ZSKIP_A);
//   * cache policies (gfx950 aux bits: 1 = sc0, 2 = nt, 16 = sc1), profiles/r04_store_policy.txt, r05_steady_policies.txt: every
//     stream is touched once per sweep -- line-aligned input blocks nt, state loads nt, state stores sc1 (what stays in the
//     Infinity Cache between the forward and the backward sweep matters), E / Ed stores nt sc1, zero-fill stores default.
constexpr int AUX_ST_STORE = 16, AUX_ST_LOAD = 2, AUX_IN_LOAD = 0, AUX_LINES_LOAD = 2, AUX_OUT_STORE = 18, AUX_ZERO_FILL = 0;
constexpr bool ABL_NOSTORE = (SDP_ABL & 1) != 0;
constexpr bool ABL_NOLOAD = (SDP_ABL & 2) != 0;
constexpr bool ABL_NOSYNC = (SDP_ABL & 4) != 0;
constexpr bool ABL_NOMATH = (SDP_ABL & 8) != 0;
constexpr bool ABL_NOLDS = (SDP_ABL & 16) != 0;  // staged inputs bypass LDS (wrong data, same dependencies)
// Progress words live in LDS and are polled by other waves.  They are accessed with explicit DS
// instructions: a volatile access through a generic pointer compiles to flat_load + vmcnt(0),
// which drains every outstanding prefetch at each poll.
__device__ __forceinline__ int lds_load_i32(unsigned addr)
{
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
// split form: issue the read, do other LDS reads behind it (a wave's DS instructions execute in order), wait once
__device__ __forceinline__ int lds_issue_i32(unsigned addr)
{
    int v;
    asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_wait(int &v)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v) : : "memory");
}
__device__ __forceinline__ void lds_store_i32(unsigned addr, int v)
{
    asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(v) : "memory");
}

template <class X>
__device__ __forceinline__ void keep(X &x)
{
    asm volatile("" : "+v"(x));
}

// ----------------------------------------------------------------------------------
// pass descriptions
// ----------------------------------------------------------------------------------
// Q formats: the forward sweep writes the weights either compact (Q_PACKED, read by the backward sweep) or as
// float2 (Q_EXACT, read by the two adjoint sweeps, whose products with the -- possibly large -- directional
// derivatives need the relative precision of fp32 for small weights as well).
enum { Q_NONE = 0, Q_PACKED = 1, Q_EXACT = 2 };

template <int PASS, bool QX>
struct Traits;
template <bool QX>
struct Traits<PASS_FWD, QX> {  // nw.py:46-62
    static constexpr int SIN = 2, SOUT = 0;
    static constexpr int QIN = Q_NONE, QOUT = QX ? Q_EXACT : Q_PACKED;
    static constexpr bool DIN = false, DOUT = false;
    static constexpr bool REV = false;
};
template <bool QX>
struct Traits<PASS_BWD, QX> {  // nw.py:120-135
    static constexpr int SIN = 0, SOUT = 1;
    static constexpr int QIN = QX ? Q_EXACT : Q_PACKED, QOUT = Q_NONE;  // QX: the training path's float2 state
    static constexpr bool DIN = false, DOUT = false;
    static constexpr bool REV = true;
};
template <bool QX>
struct Traits<PASS_AFWD, QX> {  // nw.py:178-199
    // QX = true here means: the seed Ztheta is not read but formed from the loss's operands -- three staged planes
    // (ref, pred, G) instead of (Ztheta, ZA); see "fused loss seed" in the step body
    static constexpr int SIN = QX ? 3 : 2, SOUT = 0;
    static constexpr int QIN = Q_EXACT, QOUT = Q_NONE;
    static constexpr bool DIN = false, DOUT = true;
    static constexpr bool REV = false;
};
template <bool QX>
struct Traits<PASS_ABWD, QX> {  // nw.py:251-267
    static constexpr int SIN = 1, SOUT = 1;
    static constexpr int QIN = Q_EXACT, QOUT = Q_NONE;
    static constexpr bool DIN = true, DOUT = false;
    static constexpr bool REV = true;
};

// The packed state (read by the backward sweep only; see Q_EXACT above): two 20-bit fields per cell, 5 bytes (round 4; rounds
// 1-3 kept two 24-bit fields, 6 bytes; 18-bit fields were built in round 5, gated by emulation and NOT adopted -- one per cent of
// time for three quarters of the margin; two unorm16 per cell were rejected in round 1: their rounding error is carried along an
// alignment path like a random walk and passes 1e-4 on E beyond ~1000 residues.  DESIGN.md section 2, DESIGN_HISTORY.md).
// A field is the low 20 bits of the float f = 8 + q * (1 - 2^-19): in [8, 16) one ulp is 2^-20, so the fma's own rounding puts q on
// a grid of 2^-20 (absolute error <= 2^-21 = 4.8e-7 per weight; emulated on the float64 oracle's weights,
// tools/emu_state_formats.py: max |dE| 9e-7 on the benchmark's scores, 3.4e-6 on peaked ones at 512 x 512 -- the 1e-4 bound is 30x
// away, and problems with N + M > 4096 take the exact state).  The factor keeps the field below 2^20 for q <= 1 + 9e-7 -- a weight
// computed as c / sum * u can exceed 1 by a few ulp -- so no clamp is needed; a saturated weight (anything within 2^-21 of 1)
// decodes to exactly 1, a weight below 2^-21 to exactly 0, so a saturated path loses nothing.  Four cells -- eight fields, x0 y0 x1
// y1 x2 y2 x3 y3 from bit 0 up -- fill five dwords; the 20 dwords of a 16-step block are stored as five rows of one dwordx4 per
// lane (sdp_kernels.h).  The derivative state Qd is signed and unbounded and stays float2.
constexpr float QF_SCALE = 0.99999809265136718750f;    // 1 - 2^-19
constexpr float QF_UNSCALE = 1.0000019073486328125f;   // 1 + 2^-19 = 1 / (1 - 2^-19) to fp32
constexpr float QF_BASE = 8.0f;
// raw bits fx[k], fy[k] of the four cells' biased floats (0x41000000 | field) -> five dwords
__device__ __forceinline__ void q20_pack4(const unsigned *fx, const unsigned *fy, unsigned *w)
{
    // (a left shift pushes the exponent byte out of the dword, and bits 20-23 of a biased float are zero, so only fields
    //  that stay below bit 24 after their shift need masking)
    w[0] = (fx[0] & 0xfffffu) | (fy[0] << 20);
    w[1] = __builtin_amdgcn_ubfe(fy[0], 12, 8) | (fx[1] << 8) | (fy[1] << 28);
    w[2] = __builtin_amdgcn_ubfe(fy[1], 4, 16) | (fx[2] << 16);
    w[3] = __builtin_amdgcn_ubfe(fx[2], 16, 4) | ((fy[2] & 0xfffffu) << 4) | (fx[3] << 24);
    w[4] = __builtin_amdgcn_ubfe(fx[3], 8, 12) | (fy[3] << 12);
}
__device__ __forceinline__ float q20_field(unsigned u)  // field in the low 20 bits of u, anything above -> f - 8
{
    // gfx9 allows one constant-bus operand per VALU instruction, so the compiler, given two literals ((u & mask) | bias), emits
    // two instructions; with the bias in a register the bit-field insert does it in one (15.5 -> 13.75 VALU per step).  Same bits.
    unsigned r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "s"(0xfffffu), "v"(u), "v"(0x41000000u));
    return __uint_as_float(r) - 8.0f;
}
// (f - 8) of both weights of cell `sub` (0..3) of a five-dword record; the caller multiplies by QF_UNSCALE
__device__ __forceinline__ float2 q20_unpack(const unsigned *w, int sub)
{
    switch (sub) {
    case 0: return make_float2(q20_field(w[0]), q20_field(__builtin_amdgcn_alignbit(w[1], w[0], 20)));
    case 1: return make_float2(q20_field(w[1] >> 8), q20_field(__builtin_amdgcn_alignbit(w[2], w[1], 28)));
    case 2: return make_float2(q20_field(__builtin_amdgcn_alignbit(w[3], w[2], 16)), q20_field(w[3] >> 4));
    default: return make_float2(q20_field(__builtin_amdgcn_alignbit(w[4], w[3], 24)), q20_field(w[4] >> 12));
    }
}

// Exact-state weights: the largest of the three is formed as 1 - (the other two).  c/sum*u goes through an
// approximate reciprocal and two products (~1.5 ulp): harmless for a weight of 0.3, but a weight that the reference
// -- which divides in float64 and rounds once (nw.py:21-22,115) -- stores as exactly 1.0 would come out as
// 1 +- 1e-7, and the adjoint sweeps turn such an error into (1 - q) * a, relative to a difference that should be
// 0: on long, peaked alignments it reached 1e-3 of Ed.  The small weights carry the same relative error, so the
// complement is accurate to 1e-7 * (1 - q).  When the match weight qm is the largest nothing has to be done: the
// readers form qm = 1 - qx - qy anyway.
typedef float f32x2 __attribute__((ext_vector_type(2)));  // operand of the packed fp32 instructions (v_pk_mul / v_pk_fma)

// 2^(theta log2e) = 2^tt * (1 + c) where tt = fl(theta * fl(log2e)) is what v_exp_f32 was given and
// c = ln2 * (theta * log2e - tt), the rounding of the product recovered exactly (fma) plus the low part of log2(e).
// |c| <= |tt| 2^-24: the second-order term is below 1e-12.
__device__ __forceinline__ f32x2 exp2_residual(f32x2 theta, f32x2 tt)
{
    constexpr float L_HI = 1.44269502162933349609375f, L_LO = 1.92596299112661746e-8f, LN2 = 0.69314718055994530942f;
    f32x2 d = __builtin_elementwise_fma(theta, (f32x2){L_HI, L_HI}, -tt);
    d = __builtin_elementwise_fma(theta, (f32x2){L_LO, L_LO}, d);
    return d * (f32x2){LN2, LN2};
}

__device__ __forceinline__ void q_sharpen(float &wx, float &wy, float wm)
{
    const float big = __builtin_fmaxf(wx, wy), small = __builtin_fminf(wx, wy);
    const float o = 1.0f - (small + wm);
    const float nb = big > 0.5f ? o : big;
    const bool xbig = wx >= wy;
    wx = xbig ? nb : wx;
    wy = xbig ? wy : nb;
}

// ----------------------------------------------------------------------------------
// carries
// ----------------------------------------------------------------------------------
// How a pass represents the values that flow from cell to cell (and across strips):
//   CK_F64 : float64, as the reference does internally (nw.py:49-53,125,182-185,256).
//   CK_F32 : float32.  Used by the backward sweep: E is a sum of non-negative products of weights
//            in [0,1], so there is no cancellation and fp32 accumulation stays ~1e-6 of the result.
//   CK_EXP : scaled exp-domain pair (a, e) with V = e*ln2 + ln(a), a in [0.5,1).  The forward
//            recurrence  V = theta + log(e^(A+up) + e^diag + e^(A+left))  becomes
//            alpha = c_theta * (c_A*(u + l) + d): one fma chain with no exp/log on the dependency
//            chain (the HMM "scaled forward algorithm").  theta and A are split into an integer
//            power of two (added to the exponent) and a factor in [1,2), and the three operands are
//            aligned to their largest exponent, so no finite input can overflow or cancel; every
//            rescale is an exact power of two, so the only rounding is the fp32 fma chain itself.
// (the enum and boundary_slot_bytes live in sdp_kernels.h: the host sizes the LDS rows by them.  Which pass uses which kind was a
// build-time choice until round 6; the alternatives -- a float64 forward, a float64 backward, an fp32 adjoint backward -- lost by
// measurement or by parity in rounds 1-2 and are gone.)

template <int PASS>
struct Kind {
    static constexpr int value = PASS == PASS_FWD ? CK_EXP : (PASS == PASS_BWD ? CK_F32 : CK_F64);
};

typedef unsigned long long u64;  // one boundary slot (LDS) / one edge value in registers

__device__ __forceinline__ u64 pack2(unsigned lo, unsigned hi) { return ((u64)hi << 32) | lo; }
__device__ __forceinline__ unsigned lo32(u64 x) { return (unsigned)x; }
__device__ __forceinline__ unsigned hi32(u64 x) { return (unsigned)(x >> 32); }

// V = 0 in the exp-domain representation: 0.5 * 2^1
// Windowed form of the exp-domain forward (see steps_wf in sweep): values of one chunk are plain floats relative
// to a per-lane exponent ("frame").  After the K steps every value a lane produced must lie in [WF_LO, WF_HI]:
//   * overflow anywhere in a step gives inf (or NaN), which stays in the lane's value and trips the upper test;
//   * the factors 2^theta and 2^A of every step must not exceed WF_FMAX = 2^12 (|theta|, |A| <= 8.3; anything
//     else, including NaN, goes to the normalised form).  Then a value >= WF_LO = 2^-100 proves that the sum it
//     was made from was a normal float (>= 2^-112), and values <= WF_HI = 2^110 keep every sum below 2^124, so its
//     reciprocal is a normal float too.  A = -inf (2^A = 0) is fine.
// A chunk that fails a test is redone in the per-step-normalised form; nothing was committed before the test.
// WF_HI also leaves room for the consumers of published values (value * 2^A * 2 + ... stays below 2^128).
// Values mostly grow along a row (fastest in the lower left corner of the matrix: ~5 bits per step, and 10-20 in a row's first
// columns), so a lane's frame is placed WF_BIAS bits above the exponent of its current value: it starts the block at
// 2^-WF_BIAS, with 110 + WF_BIAS bits of room upwards and 100 - WF_BIAS downwards.  Round 4: 40 -> 64.  With 40 the head
// blocks of the two bottom strips of a 512 x 512 pair failed the range test (rows 400+ gain 152-158 bits in the 16 steps
// after their first column; cycle stamps: 6 + 5 blocks at 7-10k cycles instead of 2k, all on the pair's critical path --
// strip 7 cannot start before strip 6's head is through), ~11 % of the launch.  Blocks that fail, by bias (40 / 56 / 64 /
// 80) on 512 x 512: benchmark scores 8 / 0 / 0 / 0; theta x 8: 191 / 127 / 110 / 73; theta - 2: 0 / 0 / 0 / 159; theta x 8 - 4:
// 47 / 38 / 35 / 76; A + 0.5: 10 / 0 / 0 / 0 (emulated from the float64 V).  Which form a block runs in does not change a
// cell's value, and only in the packed state whether its weights are sharpened (see norm_block).
constexpr unsigned WF_HI = 0x76800000u;  // 2^110
constexpr unsigned WF_LO = 0x0d800000u;  // 2^-100
constexpr unsigned WF_FMAX = 0x45800000u;  // 2^12
#ifndef SDP_WF_BIAS
#define SDP_WF_BIAS 64
#endif
constexpr int WF_BIAS = SDP_WF_BIAS;
constexpr int WB = 16;  // steps per frame (every chunk length is a multiple)
constexpr int FRAME_NONE = (int)0x80000000;  // published chunk is not in one frame (per-value exponents apply)
constexpr float EXP_ONE_A = 0.5f;
constexpr int EXP_ONE_E = 1;

template <int KIND>
__device__ __forceinline__ u64 edge_zero()
{
    if constexpr (KIND == CK_EXP) return pack2(__float_as_uint(EXP_ONE_A), (unsigned)EXP_ONE_E);
    else return 0ull;  // +0.0 as f64 and as f32
}

// bank-spreading permutation of the staged-input ring (see "Staged INPUT geometry"): 0,4,1,5,2,6,3,7
__device__ __forceinline__ constexpr int ring_pi(int x) { return ((x & 1) << 2) | (x >> 1); }

// per-lane recurrence state carried from step to step
struct Carry {
    // CK_F64: a = own V / value sent up, b = previous `up` / own py, c = pm of the previous step
    double a, b, c;
    // CK_F32 (reverse): same roles in fp32
    float fa, fb, fc;
    // CK_EXP (forward): own (alpha, exponent), and the diagonal predecessor's pair
    float xa;
    int xe;
    float da;
    int de;
};

// per-cell terms of the masked alignment losses (used by the loss kernels and by the fused seed of the adjoint forward)
__device__ __forceinline__ float loss_clamp(float p)
{
    const float eps = 3e-8f;  // losses.py:27
    return fminf(fmaxf(p, eps), 1.0f - eps);
}

// value of one counted cell (kind as above)
__device__ __forceinline__ float loss_term(float r, float y, int kind)
{
    if (kind == 0) {
        const float p = loss_clamp(y);
        return r * logf(p) + (1.0f - r) * logf(1.0f - p);
    }
    const float d = kind == 1 ? r * y : r - y;
    return d * d;
}
// derivative factor of one counted cell w.r.t. the predicted value
__device__ __forceinline__ float loss_dterm(float r, float y, float sc, int kind)
{
    if (kind == 0) {
        const float eps = 3e-8f;
        return (y >= eps && y <= 1.0f - eps) ? sc * (r / y - (1.0f - r) / (1.0f - y)) : 0.f;  // clamp passes the gradient inside only
    }
    return kind == 1 ? sc * r * r * y : sc * (r - y);
}

// ----------------------------------------------------------------------------------
// the sweep
// ----------------------------------------------------------------------------------
// GEN: the "general pitch" instantiation -- staged blocks are aligned to 128-byte lines of MEMORY through per-row offsets
// computed at run time (any row pitch, any plane offset).  GEN = false is the instantiation for rows and planes that
// start on K-float boundaries (M a multiple of 32, 128-byte aligned tensors): the same formulas with the offsets
// folded to constants -- the host picks per launch (sdp_api.hip), and the headline shape pays nothing for generality
// (with one instantiation for both, the forward kernel measured +5 % and the backward +3 % at M = 512).
template <int PASS, int K, bool QX = false, bool LINES = false, bool GEN = false, bool PARTS = false, bool NOPIPE = false, bool NOCLEAN = false>
__device__ __forceinline__ void sweep(const Params &p)
{
    using T = Traits<PASS, QX>;
    constexpr bool REV = T::REV;
    constexpr int KIND = Kind<PASS>::value;
    constexpr bool CLEAN = PASS == PASS_FWD && !NOCLEAN;   // forward sweep: what lies beside the matrix takes no part in anything (see need_clean)
    static_assert(!NOCLEAN || (!GEN && !PARTS), "only the aligned-pitch one-workgroup-per-pair builds have a twin without the cleaning");
    constexpr int RPI = 64 / K;    // tensor rows covered by one staged (dword) store instruction
    constexpr int PITCH = 2 * K;   // LDS pitch of a staged input plane: a ring of two K-column blocks per row
    constexpr int RING = 2 * K;
    constexpr int PLANE = 64 * PITCH;
    constexpr int LPR = K / 4;     // staged loads move 4 columns per lane: lanes per K-column block of a row,
    constexpr int RPL = 64 / LPR;  // rows per load instruction,
    constexpr int NLD = K / 4;     // load instructions per plane and chunk
    constexpr int QMAX = (63 + K - 1) / K;  // largest ceil(r/K) over the 64 rows of a strip
    constexpr int PO = stage_out_pitch(PASS, K);  // LDS pitch of the staged output ring: two chunks per row + 1 (adjoint backward sweep: 64 steps + 1, see FLUSH2)
    constexpr int NSTAGE = T::SIN + T::SOUT;
    constexpr int NS = T::SIN > 0 ? T::SIN : 1;
    constexpr int PUB_LANE = REV ? 0 : 63;    // lane that produces this strip's boundary row
    constexpr int DPP_IN = REV ? DPP_WAVE_SHL1 : DPP_WAVE_SHR1;  // pull from the lane that owns the previous row

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = blockDim.x >> 6;
    const int parts = PARTS ? p.parts : 0;   // (only the PARTS instantiations carry the bridge code: in the others it costs the forward sweep a fifth of its speed)
    // batches with per-pair lengths and more pairs than CUs are launched longest-first (p.order: pair handled by
    // each workgroup, written by sdp_order_kernel): the hardware hands workgroups to CUs in index order as they
    // free up, which then is the longest-processing-time-first rule
    //
    // PARTS (parts > 0; forward and fp32 backward sweeps): a pair's strips are cut into parts of parts strips and every
    // part is a workgroup of its own -- on its own CU -- so that a long pair is swept by several CUs at once, each with
    // one strip per wave, instead of one CU taking the strips in rounds.  The boundary between the last strip of a part
    // and the first strip of the next crosses CUs through global memory (see "bridge" below).  Which (pair, part) a
    // workgroup takes follows from a priority order in which a consumer always has a higher index than its producer:
    // workgroups are dispatched in index order, so a waiting part never keeps its producer off the chip.
    int wg_pair = (int)blockIdx.x, wg_part = 0;
    if (parts) {
        if (p.wg_map) {   // per-pair lengths: the order sdp_parts_map_kernel worked out
            const int e = p.wg_map[blockIdx.x];
            wg_pair = e / p.nparts_max, wg_part = e % p.nparts_max;
        } else {          // equal pairs: every pair's part 0, then every pair's part 1, ...
            wg_pair = (int)blockIdx.x % p.B, wg_part = (int)blockIdx.x / p.B;
        }
    }
    // (b and the part are the same for the whole workgroup, but come out of memory: say so -- without it the forward
    //  parts kernels of the shipped build took every buffer descriptor derived from them for divergent, wrapped 150-260
    //  loads and stores in readfirstlane loops and waited for each bridge load on the spot: 672 instead of 585 us at
    //  BASELINE configs[2])
    wg_pair = __builtin_amdgcn_readfirstlane(wg_pair), wg_part = __builtin_amdgcn_readfirstlane(wg_part);
    const int b = __builtin_amdgcn_readfirstlane((p.order && !parts) ? p.order[wg_pair] : wg_pair);

    int n = p.N, m = p.M;
    if (p.lens) {
        n = p.lens[2 * b];
        m = p.lens[2 * b + 1];
        n = n < 1 ? 1 : (n > p.N ? p.N : n);
        m = m < 1 ? 1 : (m > p.M ? p.M : m);
    }
    n = __builtin_amdgcn_readfirstlane(n);
    m = __builtin_amdgcn_readfirstlane(m);
    // routed launches (per-pair lengths; sdp_api.hip: exact_for): thin long pairs belong to the exact-state build's launch, all
    // others to the packed-state build's -- the workgroups of the other kind leave before they touch anything
    if (p.route != 0 && thin_pair(n, m) != (p.route == 2)) return;
    const int nstrips = (n + 63) >> 6;
    const int nchunks = (m + 63 + K - 1) / K;  // steps t in [0, m+63)
    const bool sw = p.variant == SDP_SW;
    // strips of this workgroup: [s_lo, s_hi).  Reverse sweeps count their parts from the END of the pair (part 0 = the last
    // strips, swept first), so that also there a producer has the lower workgroup index.
    const int nparts = parts ? (nstrips + parts - 1) / parts : 1;
    const int part = wg_part;
    const bool absent = part >= nparts;   // this pair has fewer parts than the longest pair of the batch: see zero_fill
    const int part_pos = (REV && parts) ? nparts - 1 - part : part;   // position of the part in strip order
    const int s_lo = parts ? part_pos * parts : 0;
    const int s_hi = parts ? (s_lo + parts < nstrips ? s_lo + parts : nstrips) : nstrips;
    const int nstrips_wg = s_hi - s_lo;

    // ---- LDS carve: boundary rows (8-byte slots), progress words, per-wave staging ----
    // Two boundary rows suffice for any number of waves: strip s+2 can only overwrite column c of the row it
    // shares with strip s after strip s+1 -- the reader of that row -- has produced its own column c.  The
    // progress words are per wave (strips that share a word are processed one after the other by that wave).
    constexpr int nslot = 2;
    // (a slot holds what one column hands down: 8 bytes -- a float64, or the forward sweep's value / exponent pair -- and
    //  4 in the fp32 backward sweep, whose LDS need at M <= 1024 thereby stays under half a CU's: two workgroups per CU)
    using slot_t = std::conditional_t<boundary_slot_bytes(PASS) == 4, unsigned, u64>;
    slot_t *bnd = reinterpret_cast<slot_t *>(smem);
    const unsigned prog = (unsigned)(uintptr_t)(bnd + (size_t)nslot * p.mcap);  // LDS byte address of word 0
    // frame words (forward sweep): one per boundary row and producer chunk, FRAME_NONE or the common exponent of
    // the K values that chunk published
    int *frm = reinterpret_cast<int *>(bnd + (size_t)nslot * p.mcap) + 16;
    float *stage = reinterpret_cast<float *>(smem + p.stage_off) + (size_t)wave * stage_floats(PASS, K, T::SIN);
    float *lds_in = stage;
    float *lds_out = stage + T::SIN * PLANE + stage_out_pad(PASS);   // (the pad: FLUSH2 writes one float in front of a row, and row 0's would be the input ring's last)

    if (threadIdx.x <= (unsigned)W) lds_store_i32(prog + 4 * threadIdx.x, 0);   // (word W: imported boundaries)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the store above is inline asm: the compiler's own counter tracking does not see it
    __syncthreads();

    // ---- per-pair tensor descriptors ----
    const size_t plane_elems = (size_t)p.N * p.M;
    const unsigned plane_bytes = (unsigned)(plane_elems * 4);

    // Lengths-aware mode: E / Ed are dense (B,N,M) tensors that must be zero outside the pair's n x m block.  The
    // sweep only writes the block; the rest is zero-filled here by the pair's own workgroup (no separate memset
    // pass over the whole tensor): by the waves that have no strip (short pairs -- they start at once and finish
    // long before the batch's longest pair), otherwise by every wave after its last strip.
    // With PARTS a pair has as many workgroup slots as the longest pair of the batch has parts; the first slot past its
    // own parts (an "absent" part: the shorter the pair, the more there is to fill) does the fill, all four waves, and
    // the parts that sweep do none.  Measured on BASELINE configs[2] (backward sweep, 784 MB of zeros next to 724 MB of
    // sweep traffic; 350 us with the fill switched off): fill by each pair's first part after its sweep 506 us; shared by
    // all absent slots 475-510; by the first absent slot 445-490 (adopted); absent slots first in the dispatch order 533;
    // throttled with s_sleep 500-870; nt / sc0 sc1 stores no change.  The fill costs about what it would cost alone.
    auto zero_fill = [&]() {
        if constexpr (T::SOUT > 0) {
            if (p.lens == nullptr || (n == p.N && m == p.M)) return;
            if (p.flags & 2) return;   // SDP_NO_FILL: the caller never reads E / Ed outside the pairs' blocks
#ifdef SDP_NO_ZERO_FILL
            return;   // timing experiment only: E keeps whatever was in the buffer outside the pair's block
#endif
            const int nabsent = parts ? p.nparts_max - nparts : 0;
            if (nabsent > 0 ? !absent : part != 0) return;
            if (absent && wg_part != nparts) return;   // (the first of them does it all: the later ones come last in the dispatch order)
            const int idle = W > nstrips_wg ? W - nstrips_wg : 0;
            const int parts = absent ? W : (idle > 0 ? idle : W);                            // waves that share the fill,
            const int part = absent ? wave : (idle > 0 ? wave - nstrips_wg : wave);          // this one's index among them
            if (part < 0) return;
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            constexpr int ZF_AUX = AUX_ZERO_FILL;
            const u32x4 z4 = {0u, 0u, 0u, 0u};
            __amdgpu_buffer_rsrc_t rz = make_rsrc(p.sout + ((SDP_EXP_BUILD && (p.dbg & 2)) ? 0 : (size_t)b) * plane_elems, plane_bytes);
            // rows [n, N): one contiguous run; stores past the end of the plane are dropped dword by dword
            const unsigned tail1 = (unsigned)(p.N * p.M);
            for (unsigned e = (unsigned)(n * p.M) + (unsigned)(part * 64 + lane) * 4u; e < tail1; e += (unsigned)parts * 256u)
                __builtin_amdgcn_raw_buffer_store_b128(z4, rz, e * 4u, 0, ZF_AUX);
            // rows [0, n): columns [m, M)
            if (m < p.M) {
                for (int r = part; r < n; r += parts) {
                    const unsigned row = (unsigned)(r * p.M);
                    for (int c = m + 4 * lane; c < p.M; c += 256) {
                        if (c + 4 <= p.M) {
                            __builtin_amdgcn_raw_buffer_store_b128(z4, rz, (row + c) * 4u, 0, ZF_AUX);
                        } else {
                            for (int q = 0; q < 4; ++q)
                                if (c + q < p.M) __builtin_amdgcn_raw_buffer_store_b32(0u, rz, (row + c + q) * 4u, 0, ZF_AUX);
                        }
                    }
                }
            }
        }
    };
    if (absent || wave >= nstrips_wg) {
        zero_fill();
        return;
    }
    // experiments build only (sdp_set_debug): all pairs alias pair 0 -- every access is served from cache
    const size_t b_in = (SDP_EXP_BUILD && (p.dbg & 1)) ? 0 : b, b_out = (SDP_EXP_BUILD && (p.dbg & 2)) ? 0 : b,
                 b_st = (SDP_EXP_BUILD && (p.dbg & 4)) ? 0 : b;
    __amdgpu_buffer_rsrc_t rs_in[NS];
    if constexpr (T::SIN > 0) {
        rs_in[0] = make_rsrc(p.sin0 + b_in * plane_elems, plane_bytes);
        if constexpr (T::SIN > 1)
            rs_in[1] = make_rsrc(p.sin1 ? p.sin1 + b_in * plane_elems : p.sin0, p.sin1 ? plane_bytes : 0u);
        if constexpr (T::SIN > 2) rs_in[2] = make_rsrc(p.sin2 + b_in * plane_elems, plane_bytes);
    }
    __amdgpu_buffer_rsrc_t rs_out = make_rsrc(T::SOUT ? (const void *)(p.sout + b_out * plane_elems) : (const void *)p.vout,
                                              T::SOUT ? plane_bytes : 0u);

    // per-lane constants of the staged-chunk geometry: lane -> (row r_l within RPI, step s_l)
    const int r_l = lane / K, s_l = lane % K;
    const int r4_l = lane / LPR, cg_l = lane % LPR;  // staged loads: lane -> (row within RPL, 4-column group)
    const int ld = p.M;
    const int lane_off = (r_l * ld + s_l - r_l) * 4;   // byte offset of this lane's element for k = 0, i0 = t0 = 0

    const float et = (PASS == PASS_BWD) ? p.vin[p.vin_bcast ? 0 : b] : 0.f;
    const float seed_scale = (PASS == PASS_AFWD && QX) ? p.vin[b] : 0.f;   // fused loss seed: per-pair factor of dLoss/dE

    for (int sidx = wave; sidx < nstrips_wg; sidx += W) {
        const int s = REV ? s_hi - 1 - sidx : s_lo + sidx;  // strip handled now
        const int i0 = s << 6;
        const int rows = (n - i0) < 64 ? (n - i0) : 64;
        const bool has_pred = REV ? (s + 1 < nstrips) : (s > 0);   // strip whose boundary we consume
        const bool has_succ = REV ? (s > 0) : (s + 1 < nstrips);   // strip that consumes ours
        const int pidx = sidx - 1;                                  // producer's position in processing order
        // bridged: the strip whose boundary we consume / that consumes ours belongs to another workgroup (parts)
        const bool imported = has_pred && sidx == 0 && parts != 0;
        const bool exported = has_succ && sidx == nstrips_wg - 1 && parts != 0;
        // the boundary row an imported boundary is replayed into is the one strip sidx + 1 publishes to: that strip
        // trails this one by a whole lag and never reaches a column this strip still has to read; its progress word
        // is word W (no wave's own)
        const int pslot = has_pred ? (imported ? 1 : pidx % nslot) : 0, oslot = sidx % nslot;                 // boundary rows
        const int pword = has_pred ? (imported ? W : pidx % W) : 0, pbase = (has_pred && !imported) ? (pidx / W) * PROG_STRIDE : 0;  // progress words
        const int oword = sidx % W, obase = (sidx / W) * PROG_STRIDE;
        const slot_t *bnd_in = bnd + (size_t)pslot * p.mcap;
        slot_t *bnd_out = bnd + (size_t)oslot * p.mcap;
        const int *frm_in = frm + pslot * FRAME_CAP;
        int *frm_out = frm + oslot * FRAME_CAP;
        int wf_skip = 0;  // blocks to leave to the normalised form after a failed windowed attempt
        // chunks in which every lane sits on a real, non-special cell need no masking at all
        // (the terminal cell of the last strip is met at t >= m-1, which interior chunks never contain)
        const bool plain_strip = rows == 64 && !(sw && s == 0);

        // reverse sweeps: this lane's cell is live (inside the matrix, not on the Smith-Waterman border) at step t
        // iff live_lo <= t < live_lo + live_span
        const int live_lo = lane + (sw ? 1 : 0);
        const unsigned live_span = (lane < rows && !(sw && i0 + lane == 0)) ? (unsigned)(m - (sw ? 1 : 0)) : 0u;
        // step at which this lane meets the terminal cell (n-1, m-1) of the pair; -1 if never
        const int t_final = (s == nstrips - 1 && lane == rows - 1) ? (m - 1 + lane) : -1;

        // Skewed state addressing.  The state of a (pair, strip) is a sequence of UNITS of 32 steps; a unit is contiguous
        // (packed Q: ten rows of 1024 B, see q20_soff below; float2 states: 32 rows of 512 B, step t, lane l at t*512 + l*8).  Where
        // unit u of (pair b, strip s) lives is given by two strides the host picks (sdp_api.hip::state_layout):
        //     byte offset = (b * nstrips + s) * ps + u * us + (offset inside the unit)
        // with ps = units per strip * unit bytes, us = unit bytes: every strip is one stream.  (A "marching" layout, unit u of all
        // strips of all pairs in one slab, measured no different in round 3 and is gone; the strides stay parameters.)
        // One buffer descriptor per (pair, strip); the unit part of the offset is uniform and rides in the scalar
        // offset operand (which the hardware does not range-check), the lane offset is a per-lane constant.
        // (The descriptors' size is the largest one for which the out-of-range offset OOB still is out of range: whether or
        // not the hardware adds the scalar offset before its range check, every state access below is accepted -- and
        // nothing here relies on the check.)
        constexpr unsigned ST_RECORDS = 0x7fffffffu;
        const size_t ps_idx = b_st * p.nstrips_max + s;
        const unsigned q_lane = lane * 16;
        // (Two ways of not moving the skew padding -- the records of lanes that lie outside the matrix -- were built and dropped:
        //  per-lane skipping in round 4 (partial 128-byte lines at the edge of a ramp: forward 210 -> 215 us), whole dead lines in
        //  round 5 (bit-identical; forward alone 193 -> 185 us, nothing in the forward;backward sequence).  DESIGN_HISTORY.md.)
        __amdgpu_buffer_rsrc_t rs_q = make_rsrc(T::QIN == Q_PACKED ? (const void *)(reinterpret_cast<const char *>(p.qin) + ps_idx * p.st_ps)
                                                : (T::QOUT == Q_PACKED ? (const void *)(static_cast<char *>(p.dout) + ps_idx * p.st_ps) : (const void *)p.vout),
                                                (T::QIN == Q_PACKED || T::QOUT == Q_PACKED) ? ST_RECORDS : 0u);
        // Packed state: a 16-step BLOCK of a lane is 20 dwords (four records of five), kept as five rows of 1024 B -- row j holds
        // dwords 4j .. 4j+3 of every lane -- so that every access is a whole dwordx4 of contiguous 1 KB per wave: five memory
        // instructions per 16 steps.  A 32-step unit is two blocks, ten rows.  Scalar offset of row jr (0..4) of the block that
        // holds step t:
        auto q20_soff = [&](int t, int jr) { return (unsigned)(t >> 5) * p.st_us + (unsigned)(((t >> 4) & 1) * 5120 + jr * 1024); };
        typedef unsigned u32x4q __attribute__((ext_vector_type(4)));
        // The ring of prefetched rows keeps every dwordx4 AS THE VECTOR it was loaded as.  Kept as scalars (rounds 1-3 did
        // that for the dwordx3 records) each dword is a loop-carried value of its own, the load needs consecutive registers
        // for them, and the compiler resolved that by loading elsewhere and COPYING all rows of the next chunk into place at
        // the end of every iteration -- behind an `s_waitcnt vmcnt(0)`.
        constexpr int QROWS = 5 * K / 16;   // rows (dwordx4 per lane) of a chunk
        u32x4q rq2[2][QROWS];   // two sets: the current chunk's rows and the next chunk's (roles swap every chunk, see ROT)
        // ... and the rows of the NEXT chunk are a second set, loaded in one burst at the top of the iteration (TOPLOAD) and
        // moved over at its end: the wait for them then sits a whole iteration behind their issue.  (Refilling a slot of the
        // first set right after its last use -- the scheme of rounds 1-3 -- reads as the same thing, but the compiler gave the
        // refills registers of their own anyway and moved them over at the end of the iteration, behind a vmcnt(0) that the
        // loads issued during the last steps had had a few hundred cycles to meet: the reverse sweeps ran at memory latency.)
        constexpr bool TOPLOAD = REV && T::QIN == Q_PACKED;
        auto load_row = [&](__amdgpu_buffer_rsrc_t rs, int t_base, int jj, u32x4q &dst) {   // row jj of the chunk (five per 16-step block)
            dst = __builtin_amdgcn_raw_buffer_load_b128(rs, q_lane, q20_soff(t_base + 16 * (jj / 5), jj % 5), AUX_ST_LOAD);
        };
        auto load_q20 = [&](int t_base, int jj, auto set_tag) {   // row jj (0 .. QROWS-1) of the chunk that starts at step t_base -> set S
            load_row(rs_q, t_base, jj, rq2[decltype(set_tag)::value][jj]);
        };
        // the five dwords of the record of steps 4g .. 4g+3 of the chunk, out of the rows of set S
        auto q20_record = [&](int g, unsigned *w, auto set_tag) {
#pragma unroll
            for (int e = 0; e < 5; ++e) {
                const int d = 5 * (g % 4) + e;           // dword of the block
                w[e] = rq2[decltype(set_tag)::value][5 * (g / 4) + (d >> 2)][d & 3];
            }
        };
        auto store_q = [&](int t_base, int g, const unsigned *src) {   // g: row of the block that holds step t_base (src: its four dwords)
            // A VALU instruction that overwrites a data register of a store wider than 64 bits in the very next
            // issue slot corrupts the stored value for part of the wave on gfx950 (seen: lanes 12-15 of every
            // 16).  The compiler only inserts the wait state for stores without a scalar offset register, so it
            // is forced here: the no-op "reads" the data registers (nothing that overwrites them can move above
            // it) and is ordered after the store as a memory operation.
            u32x4q v;
            v[0] = src[0], v[1] = src[1], v[2] = src[2], v[3] = src[3];
            __builtin_amdgcn_raw_buffer_store_b128(v, rs_q, q_lane, q20_soff(t_base, g), AUX_ST_STORE);
            asm volatile("s_nop 1" : : "v"(v) : "memory");
        };
        // float2 states (Qd, and Q in its exact form).  Cells outside the matrix are stored like any other (their
        // values are never used: every reader masks them).
        const unsigned st_lane = lane * 8;
        __amdgpu_buffer_rsrc_t rs_d = make_rsrc(T::DIN ? (const void *)(reinterpret_cast<const char *>(p.din) + ps_idx * p.st2_ps)
                                                       : (T::DOUT ? (const void *)(static_cast<char *>(p.dout) + ps_idx * p.st2_ps) : (const void *)p.vout),
                                                (T::DIN || T::DOUT) ? ST_RECORDS : 0u);
        __amdgpu_buffer_rsrc_t rs_qx = make_rsrc(T::QIN == Q_EXACT ? (const void *)(reinterpret_cast<const char *>(p.qin) + ps_idx * p.st2_ps)
                                                 : (T::QOUT == Q_EXACT ? (const void *)(static_cast<char *>(p.dout) + ps_idx * p.st2_ps) : (const void *)p.vout),
                                                 (T::QIN == Q_EXACT || T::QOUT == Q_EXACT) ? ST_RECORDS : 0u);
        // scalar offset of row t_base + k8 (k8 = 0, 8, 16, 24)
        auto f2_soff = [&](int t_base, int k8) { return (unsigned)(t_base >> 5) * p.st2_us + (unsigned)(((t_base & 31) + k8) * 512); };
        auto load_f2 = [&](__amdgpu_buffer_rsrc_t rs, int t_base, int k) {  // row t_base + k
            const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs, st_lane + (k & 7) * 512, f2_soff(t_base, k & ~7), AUX_ST_LOAD);
            // NB: copy the elements to scalars first -- __builtin_bit_cast applied directly to a vector
            // element lvalue (v[1]) reads element 0 with this compiler.
            const unsigned lo = v[0], hi = v[1];
            return make_float2(__uint_as_float(lo), __uint_as_float(hi));
        };
        auto store_f2 = [&](__amdgpu_buffer_rsrc_t rs, int t_base, int k, float2 qq) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            u32x2 v;
            v[0] = __float_as_uint(qq.x);
            v[1] = __float_as_uint(qq.y);
            __builtin_amdgcn_raw_buffer_store_b64(v, rs, st_lane + (k & 7) * 512, f2_soff(t_base, k & ~7), AUX_ST_STORE);
        };
        auto load_d = [&](int t_base, int k) { return load_f2(rs_d, t_base, k); };
        // packed state, forward: the two biased fields of a cell (bits of QF_BASE + q * QF_SCALE) arrive per step; every fourth
        // step 13 shift / mask / or instructions assemble the 20-byte record of four cells, and a row of the block (four dwords per
        // lane) leaves as soon as it is complete
        unsigned qb_x[3] = {0, 0, 0}, qb_y[3] = {0, 0, 0};
        unsigned qblk[20];   // the dwords of the current 16-step block
        auto store_state_bits = [&](int t_base, int k, unsigned fx, unsigned fy) {
            if ((k & 3) == 0) qb_x[0] = fx, qb_y[0] = fy;
            else if ((k & 3) == 1) qb_x[1] = fx, qb_y[1] = fy;
            else if ((k & 3) == 2) qb_x[2] = fx, qb_y[2] = fy;
            else {
                const unsigned ax[4] = {qb_x[0], qb_x[1], qb_x[2], fx}, ay[4] = {qb_y[0], qb_y[1], qb_y[2], fy};
                const int g = (k >> 2) & 3;   // record of the block
                q20_pack4(ax, ay, qblk + 5 * g);
                const int tb16 = t_base + (k & ~15);
                if (g == 0) store_q(tb16, 0, qblk);
                else if (g == 1) store_q(tb16, 1, qblk + 4);
                else if (g == 2) store_q(tb16, 2, qblk + 8);
                else store_q(tb16, 3, qblk + 12), store_q(tb16, 4, qblk + 16);
            }
        };
        // the state this pass produces, step t_base + k
        auto store_state = [&](int t_base, int k, float2 qq) {
            if constexpr (T::QOUT == Q_PACKED) {
                store_state_bits(t_base, k, __float_as_uint(__builtin_fmaf(qq.x, QF_SCALE, QF_BASE)), __float_as_uint(__builtin_fmaf(qq.y, QF_SCALE, QF_BASE)));
            } else if constexpr (T::QOUT == Q_EXACT) {
                store_f2(rs_qx, t_base, k, qq);
            } else {
                store_f2(rs_d, t_base, k, qq);
            }
        };

        Carry cy;
        cy.a = cy.b = cy.c = 0.0;
        cy.fa = cy.fb = cy.fc = 0.f;
        cy.xa = cy.da = EXP_ONE_A;
        cy.xe = cy.de = EXP_ONE_E;
        u64 vt_keep = edge_zero<KIND>();  // fwd passes: terminal cell's value, captured when this lane reaches it

        // Exact zeros in the ADJOINT backward sweep (round 5, ZSKIP_A).  Where E is exactly zero -- 47 % of its cells on the
        // benchmark's scores, see ZSKIP -- the adjoint backward recurrence ed = in + b, (a, b, c) = (dx e + qx ed + c', dy e + qy ed,
        // dm e + qm ed) with e = E = 0 computes nothing but zeros as long as what reaches the chunk is zero: Ed = 0 there.  Such a
        // chunk is not run, and its Q and Qd rows (16 of the sweep's 28 bytes per cell) are not read.  Whether E is zero over a
        // chunk must be known BEFORE that chunk's rows are requested, a chunk ahead: the staged E blocks therefore travel two
        // chunks ahead of the sweep instead of one (two register sets; an arriving block set is tested with one ballot), and a
        // chunk's E is zero if the two block sets that cover it are.  Zeros with a sign: a computed zero chunk leaves +0 in
        // a and b but may leave -0 in c (qm < 0 by a rounding, times +0); the skipped one leaves +0.  The difference can only
        // ever surface as the SIGN of a zero in Ed, so Ed is stored as (float)ed + 0.0f in both paths: bit-identical results.
        constexpr bool ZSKIP_A = PASS == PASS_ABWD && !ABL_NOMATH && !ABL_NOLOAD;
        float rs[ZSKIP_A ? 2 : 1][NS][K];   // staged inputs of the NEXT chunk (registers); ZSKIP_A: of the next two chunks
        // exact Q rows / Qd rows: slot k of a set holds step t0+k of a chunk; two sets whose roles (current chunk / next chunk)
        // swap every chunk (ROT)
        float2 rqx2[2][K];
        float2 rdd2[2][K];
        // TOPLOAD_X: the rows of the next chunk are a second set, loaded in one burst at the top of the iteration (see TOPLOAD)
        constexpr bool TOPLOAD_X = T::QIN == Q_EXACT || T::DIN;
        // fp32 reverse sweep: the K boundary values of a chunk travel through LDS as 16-byte accesses (K / 4 instead of K instructions each way)
        constexpr bool VEC_BND = PASS == PASS_BWD && sizeof(slot_t) == 4 && K % 4 == 0;
        // Exact zeros (see the chunk loop): the fp32 backward sweep skips the steps of chunks that can only produce +0, and (LAZY)
        // does not fetch the state rows of a chunk it already knows to be one.  There is ONE place per iteration where the next
        // chunk's rows are requested, and the request is always issued: a fetch that is not wanted goes through a buffer
        // descriptor of zero records, which returns zeros without touching memory.  (Loads under a branch, or at a second site
        // for the rare case, make the loaded registers phi nodes: copies behind waits, and a conservative wait in front of the
        // rows' first use -- measured, both.)
        constexpr bool ZSKIP = PASS == PASS_BWD && !ABL_NOMATH;
        constexpr bool LAZY = ZSKIP && !ABL_NOLOAD;
        bool known_zero = false;   // the chunk about to be processed is a zero chunk (found out an iteration ahead): its rows were not fetched
        bool za_b1 = false, za_b2 = false;   // (ZSKIP_A) the staged E block sets c and c + 1 -- the two that cover the chunk about to be processed -- are all zero
        int zring = 0;             // bit h: half h of the output ring is known to hold +0 everywhere (written by a zero chunk)
        int tail_st = 0;           // (PIPE) memory instructions this wave issued BEHIND its latest request for state rows: the stores of a pipelined chunk
        auto load_rows = [&](int tn, auto set_tag, bool wanted) {   // state rows of the chunk that starts at step tn -> set S
            if constexpr (TOPLOAD) {
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(reinterpret_cast<const char *>(p.qin) + ps_idx * p.st_ps, wanted ? ST_RECORDS : 0u);
#pragma unroll
                for (int jj = 0; jj < QROWS; ++jj) load_row(rs, tn, jj, rq2[decltype(set_tag)::value][jj]);
            } else if constexpr (T::QIN == Q_EXACT) {
                const __amdgpu_buffer_rsrc_t rs = make_rsrc(reinterpret_cast<const char *>(p.qin) + ps_idx * p.st2_ps, wanted ? ST_RECORDS : 0u);
#pragma unroll
                for (int k = 0; k < K; ++k) rqx2[decltype(set_tag)::value][k] = load_f2(rs, tn, k);
            }
        };
        // LAZY: will the chunk that starts at step tn meet nothing but +0 from outside -- the K boundary values of the strip
        // below and, where it holds the terminal cell, the cotangent?  (With +0 carries that makes it a zero chunk.)  Asked an
        // iteration ahead, so only if the strip below has published those columns already (`block`: wait for it -- the strip's
        // first chunk, which has to wait for them anyway); false = not known.  Chunks that reach over the matrix's edges and
        // strips whose boundary comes through the bridge from another workgroup are left to find out when their turn comes.
        auto boundary_zero = [&](int tn, bool block) -> bool {
            if constexpr (LAZY) {
                if ((p.flags & 1) || (SDP_EXP_BUILD && (p.dbg & 4096))) return false;
                unsigned any = (t_final >= tn && t_final < tn + K) ? __float_as_uint(et) : 0u;
                if (has_pred) {
                    const int c_lo_n = tn - 63;
                    if (imported) return false;
                    const bool inner = c_lo_n >= 0 && c_lo_n + K <= m;
                    const int need = (c_lo_n + K > 0 && c_lo_n < m) ? m - (c_lo_n < 0 ? 0 : c_lo_n) : 0;   // (as in the chunk's own acquire)
                    bool ready = need == 0;
                    for (int spin = 0; !ready && spin < (block ? (1 << 21) : 1); ++spin) {
                        if (__builtin_amdgcn_readfirstlane(lds_load_i32(prog + 4 * pword)) >= pbase + need) {
                            ready = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (!ready) return false;   // (a hand-off that never comes is reported by the chunk's own acquire)
                    if (!inner) {   // over an edge of the matrix (the first and the last chunks of a strip): columns that exist, one by one
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            const int col = c_lo_n + k;
                            if (col >= 0 && col < m) any |= (unsigned)bnd_in[col];
                        }
                    } else if constexpr (VEC_BND) {
                        const uint4 *src = reinterpret_cast<const uint4 *>(bnd_in + (c_lo_n - 1));   // columns tn - 64 .. tn - 33: one more than needed
                        const uint4 v0 = src[0];
                        any |= v0.y | v0.z | v0.w;
#pragma unroll
                        for (int g = 1; g < K / 4; ++g) {
                            const uint4 v = src[g];
                            any |= v.x | v.y | v.z | v.w;
                        }
                        any |= (unsigned)bnd_in[c_lo_n + K - 1];
                    } else {
#pragma unroll
                        for (int k = 0; k < K; ++k) any |= (unsigned)bnd_in[c_lo_n + k];
                    }
                }
                return __builtin_amdgcn_ballot_w64(any != 0) == 0;
            } else {
                return false;
            }
        };


        // Staged INPUT geometry.  Row-major tensors enter as K-column blocks, four columns (one dwordx4) per
        // lane; row r's blocks start at columns K*j - (r mod 4).  During chunk c (steps cK .. cK+K-1) row r needs
        // columns cK-r .. cK-r+K-1, which lie in its blocks c-q and c-q+1 with q = ceil(4*floor(r/4)/K); each row
        // keeps exactly those two blocks in an LDS ring of 2K columns, and one new block per row per chunk is
        // prefetched into registers a chunk ahead ("block set" bb = block bb-q of every row).
        // Ring position of column j of row r: (j + r + 4*pi(r mod 8)) mod 2K.  The "+r" cancels the skew -- lane
        // l finds step t at position (t + 4*pi(l mod 8)) mod 2K -- and together with the (r mod 4) shift of the
        // blocks it makes every loaded dwordx4 land on one aligned 16-byte slot: K/4 ds_write_b128 per plane on
        // the way in, K/4 ds_read_b128 per plane on the way out, addresses differing between lanes by constants.
        // pi = (0,4,1,5,2,6,3,7) spreads both access patterns over the banks.
        //   li_voff : global byte offset of this lane's 4 columns of block set 0 (row i0)
        //   li_w    : LDS index they are written to when bb is even (odd: the other half, ^K)
        // LINES = true (throughput builds): the blocks start at multiples of K instead, so that every load moves
        // whole 128-byte lines (a shifted block straddles two lines, and at one pair per CU the second touch of a
        // line, one chunk later, no longer hits in L2: fabric reads were 1.8x the tensors' size); the (r mod 4)
        // rotation then happens on the way into LDS, as four dword writes per loaded dwordx4.
        // "Whole lines" means lines of MEMORY: for a row pitch that is not a multiple of K floats (or a plane that
        // does not start on a line) row r's blocks are moved left by delta_r = (r M + beta) mod K columns, so that
        // they still start on 128-byte boundaries; q becomes ceil((r - delta_r) / K), the ring positions follow the
        // columns as before (round 2: forward sweep at M = 516 was 1.5x slower than at 512 without this).
        bool li_unaligned = false;
        unsigned li_voff[NLD];
        int li_col[NLD];  // first column of this lane's group in block set 0 (non-plain path: skip groups outside the row)
        int li_w[NLD][LINES ? 4 : 1];
        if constexpr (T::SIN > 0) {
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                // rows of load instruction i.  Latency builds: 8 consecutive rows.  LINES: the rows r with r mod 4 == i mod 4 of
                // one half of the strip -- a loaded group of four columns lands at ring positions p0 .. p0 + 3 with p0 = r mod 4
                // (mod 4), so with this assignment every lane of an instruction cuts its group the same way, known at compile
                // time, into the largest ALIGNED pieces: one 16-byte write (r mod 4 = 0), two 8-byte writes (2), or 4 + 8 + 4
                // bytes (1, 3) -- 36 LDS writes per chunk instead of 64 dword writes.  (An LDS instruction costs a wave ~17
                // issue cycles whatever its width -- round 4 cycle stamps: the 64 dword writes of a chunk took 1100 cycles
                // of the ~8000 a chunk takes.)
                const int r = LINES ? ((i & 3) + 32 * (i >> 2) + 4 * r4_l) : (i * RPL + r4_l);
                static_assert(!LINES || (K == 32 && NLD == 8 && RPL == 8), "row assignment of the line-aligned staging is written for K = 32");
                if constexpr (LINES) {
                    const int beta = GEN ? (int)(((uintptr_t)(p.sin0 + b_in * plane_elems) >> 2) & (uintptr_t)(K - 1)) : 0;
                    li_unaligned = GEN && (beta != 0 || (ld & (K - 1)) != 0);
                    int delta = GEN ? ((r * ld + beta) & (K - 1)) : 0;   // (i0 * M is a multiple of K: strips are 64 rows)
                    // A group must not straddle the START of the pair's plane: its byte offset would be negative, and a
                    // negative offset fails the range check for all four dwords (the per-dword check does not wrap) -- the
                    // first columns of row 0 would read as zeros.  The few rows that begin within 3 floats of the plane
                    // start therefore move by a multiple of 4 columns only (their groups then start at column <= -4 or
                    // >= 0); their blocks begin up to 3 floats off a line boundary, which costs nothing measurable.
                    if ((i0 + r) * ld < 4) delta &= ~3;
                    const int q = (r - delta + K - 1) / K;          // >= 0: delta <= K - 1
                    li_col[i] = -K * q - delta + 4 * cg_l;
                    li_voff[i] = (unsigned)((r * ld + li_col[i]) * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        li_w[i][j] = r * PITCH + ((4 * cg_l + j - delta + RING + K * (q & 1) + r + 4 * ring_pi(r & 7)) & (RING - 1));
                } else {
                    const int q = ((r & ~3) + K - 1) / K;
                    li_voff[i] = (unsigned)((r * ld - K * q - (r & 3) + 4 * cg_l) * 4);
                    li_col[i] = -K * q - (r & 3) + 4 * cg_l;
                    li_w[i][0] = r * PITCH + ((4 * cg_l + K * (q & 1) + (r & ~3) + 4 * ring_pi(r & 7)) & (RING - 1));
                }
            }
        }

        // Staged OUTPUT geometry.  Results are written to LDS by step ([lane][step mod 2K]) and leave as K-element
        // blocks that are aligned to K floats IN MEMORY (full 128-B lines for K = 32), whatever M is: row r's blocks
        // start where (address of the row's first element + column) is a multiple of K floats, i.e. after chunk t0 the
        // row owns the complete block starting at column t0 - D_r with D_r = r - rho_r, rho_r = (r (1 - M) - beta) mod K
        // (beta: misalignment of the pair's plane); its element e was produced at step offset s = rho_r + e, i.e. in
        // this chunk (s < K) or in the previously processed one (s >= K).  For M a multiple of K and an aligned plane
        // this is rho_r = r mod K, D_r = K floor(r/K): blocks aligned to K columns.  (With blocks aligned to columns
        // only, a row pitch that is not a multiple of 32 floats made every 128-byte store straddle two lines: the
        // backward sweep ran 2.1-2.4x slower at M = 516 than at M = 512.)  fo_off0 is the LDS index when the current
        // chunk has parity 0, fo_dk the change when it has parity 1, fo_voff the global byte offset (row i0, t0 = 0).
        constexpr bool FLUSH2 = T::SOUT > 0 && !GEN && (K == 32 || PASS == PASS_ABWD);   // (see below)
        // KF: columns of a row that leave together.  The adjoint backward sweep (K = 16: its float64 rows a chunk ahead are 128
        // registers) flushes 32-column blocks all the same -- whole 128-byte lines, every second chunk, out of a ring of 64 steps
        // per row: round 6's stamps (tools/adj_trace.py) showed its 16 dword reads + 16 masked dword stores of 64-byte half
        // lines per chunk, with three 16-entry index arrays held across the loop, as 2700-3700 of a chunk's 10 600 cycles.
        constexpr int KF = FLUSH2 ? 32 : K;
        static_assert(KF % K == 0 && PO >= 2 * KF + 1, "the output ring holds two flush blocks per row");
        // (GENFAST: the general-pitch K = 32 builds, whose plain chunks leave through a path of their own -- see flush_out -- and whose
        //  rare masked flushes form these indices on the spot: three 32-entry arrays held across the chunk loop are 96 registers)
        constexpr bool GENFAST = GEN && K == 32 && T::SOUT > 0;
        constexpr bool FO_ARRAYS = T::SOUT > 0 && !FLUSH2 && !GENFAST;
        int fo_off0[FO_ARRAYS ? K : 1], fo_dk[FO_ARRAYS ? K : 1];
        unsigned fo_voff[FO_ARRAYS ? K : 1];
        bool fo_need_tail = false;
        const int fo_beta = (GEN && T::SOUT > 0) ? (int)(((uintptr_t)(p.sout + b_out * plane_elems) >> 2) & (uintptr_t)(K - 1)) : 0;
        if constexpr (T::SOUT > 0 && !FLUSH2) fo_need_tail = GEN && (fo_beta != 0 || (ld & (K - 1)) != 0);
        if constexpr (FO_ARRAYS) {
            const int beta = fo_beta;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int r = k * RPI + r_l;
                const int rho = GEN ? ((r * (1 - ld) - beta) & (K - 1)) : (r & (K - 1));   // (i0 * M is a multiple of K: strips are 64 rows)
                const int sfull = rho + s_l;
                const bool prev = sfull >= K;
                fo_off0[k] = r * PO + (prev ? sfull - K : sfull) + (prev ? K : 0);
                fo_dk[k] = prev ? -K : K;
                fo_voff[k] = (unsigned)((r * ld - (r - rho) + s_l) * 4);   // column s_l - D_r of row r
            }
        }

        // Plain flushes of the aligned K = 32 builds move TWO columns per lane (FLUSH2): 16 ds_read_b64 + 16
        // buffer_store_dwordx2 per chunk instead of 32 + 32 dword operations.  A wave can have 63 memory instructions in flight
        // (vmcnt is 6 bits); with 32 dword stores of 256 bytes next to the 16 state loads a chunk is 48 of them, little more
        // than one chunk's worth of bytes ahead of the sweep, and cycle stamps (tools/bwd_trace.py) showed the wave stalling at
        // its next memory instruction once real memory latency applied (everything cache-served: 124 us, real: 160).
        // Row r's block is elements rho_r + e of the ring, e = 0..31; with PO = 65 the pair (e, e + 1), e even, is 8-byte aligned
        // in LDS (r PO + rho_r = 66 r - 32 floor(r / 32)).  A pair never straddles the two halves of the ring when the current
        // chunk has parity 0; with parity 1 the pair rho_r + e = 31 would be ring positions 63 and 0, so every chunk also
        // writes its step 31 to position -1 of the row (the unused pad of the row above), and that pair is read from there.
        // Lanes 16 a .. 16 a + 15 of an instruction take rows x + 8 (a >> 1) + 16 (a & 1): the two rows of a 32-lane LDS group
        // are 16 apart, i.e. 32 banks apart.
        // Nothing of this geometry is kept in per-instruction registers (rounds 1-3 held three 32-entry index arrays across
        // the chunk loop): a lane's LDS index and global offset are one per-lane base each plus a constant of the instruction.
        // A wave's VALU sees 256 registers; whatever lives beyond that sits in AGPRs, and the prefetched state records of the
        // next chunk were what the compiler put there -- to be copied back at the end of every iteration behind a vmcnt(0).
        //   instruction k2 (0..15), lane -> row r = c + rl, c = (k2 & 7) + 32 (k2 >> 3), rl = 8 (a >> 1) + 16 (a & 1), a = lane >> 4;
        //   columns e_l, e_l + 1 of the row's block, e_l = 2 (lane & 15);  rho_r = r & 31 = (k2 & 7) + rl;  D_r = r - rho_r = 32 (k2 >> 3)
        //   LDS index (parity 0) = r PO + rho_r + e_l = [c PO + (k2 & 7)] + f2_l,   f2_l = rl PO + rl + e_l
        //   global float offset  = r ld - D_r + e_l   = [c ld - 32 (k2 >> 3)] + f2_g, f2_g = rl ld + e_l
        const int f2_rl = 8 * ((lane >> 4) >> 1) + 16 * ((lane >> 4) & 1), f2_el = 2 * (lane & 15);
        const int f2_l = f2_rl * PO + f2_rl + f2_el, f2_g = f2_rl * ld + f2_el;
        const int f2_rl_c = f2_rl, f2_el_c = f2_el, f2_l_c = f2_l, f2_g_c = f2_g, r_l_c = lane / KF, s_l_c = lane % KF;   // (for the opaque copies in flush_out's rare paths)
        // Plain flushes of the builds without the pipelined chunk: four columns per lane and store (the same 16 aligned 8-byte LDS reads, but 8 dwordx4 stores instead of 16 dwordx2).  Instruction k4 (0..7), lane -> row r = c + rl, c = 4 (k4 & 3) + 32 (k4 >> 2),
        //   rl = (a & 1) + 16 ((a >> 1) & 1) + 2 (a >> 2), a = lane >> 3 -- the four rows of a 32-lane LDS group are r, r + 1 (bank bases two
        //   floats apart: their 8-byte reads at columns 4 g interleave) and r + 16, r + 17 (32 banks further); columns e_l .. e_l + 3, e_l = 4 (lane & 7)
        //   LDS index (parity 0) of pair j = [c PO + 4 (k4 & 3)] + f4_l + 2 j;  global float offset = [c ld - 32 (k4 >> 2)] + f4_g
        const int f4_a = lane >> 3, f4_rl = (f4_a & 1) + 16 * ((f4_a >> 1) & 1) + 2 * (f4_a >> 2), f4_el = 4 * (lane & 7);
        const int f4_l = f4_rl * PO + f4_rl + f4_el, f4_g = f4_rl * ld + f4_el, f4_s = f4_rl + f4_el;

        // Chunk c of this strip touches only real cells of a full, unmasked strip: no masking needed.
        auto chunk_interior = [&](int c) { return plain_strip && c * K >= 63 && c * K + K < m; };

        // every address of block set bb in range: the uniform part may ride in the scalar offset (no VALU
        // add and no reliance on how the hardware range-checks the scalar offset)
        // (blocks moved left by up to K-1 columns: one block set later)
        // (the latency builds' groups start at columns K j - (r mod 4): block set QMAX still holds groups that straddle column 0 of
        //  the rows with q = QMAX -- up to three floats of the row above -- so there, too, plain begins one block set later)
        auto block_plain = [&](int bb) { return rows == 64 && bb >= QMAX + ((li_unaligned || (CLEAN && !LINES && p.lens != nullptr)) ? 1 : 0) && (bb + 1) * K <= m; };
        // Forward sweep: what lies BESIDE the matrix must not take part in anything (round 6).  A loaded group of four columns may
        // straddle column 0 or column m -- up to three floats of the neighbouring row, of the neighbouring pair, of the padding of a
        // batch with per-pair lengths, or of whatever follows the tensor in memory -- and the rows below a partial strip are rows of
        // another problem altogether.  No RESULT ever depended on them (cells outside the matrix hand nothing to cells inside), but
        // the windowed form's range test looks at every lane's inputs and values, so WHICH FORM a block ran in did -- and with it
        // the timing, and in the packed state the last bits (the per-step form sharpens its weights): round 5's wrong-result bug
        // needed a plane offset of 3 floats to show for exactly this reason.  Now: rows below the strip's last row are not fetched
        // (out-of-range offset: zeros), and elements outside [0, m) are replaced by 0 on their way into the LDS ring, so that every
        // cell outside the matrix computes from theta = A = 0.  Only block sets that touch an edge pay (the non-plain ones), and
        // only where something foreign can be met at all: per-pair lengths, partial strips, unaligned pitch (GEN), the latency
        // builds' shifted groups -- the aligned throughput build on full strips without lengths loads whole groups inside or
        // outside a row and nothing else (tests/test_robustness_gpu.py::test_what_lies_beside_the_matrix_...).
        // What it costs, and who pays (steady state, same box, main vs the same build without any of this: forward sweep of
        // 256 x 512^2 169.6 vs 166.2 us although nothing was cleaned there -- the mere presence of the code costs registers in the
        // ramp chunks, which are a pair's critical path; 64 x 512^2 on the latency build, cleaning every edge block set: 143.9 vs
        // 135.2, and still 140.0 with the code present but never executed).  So: CLEAN is a property of the BUILD.  The aligned-pitch
        // builds exist twice -- without a trace of it (NOCLEAN: sdp_fwd_kernel, sdp_fwd_x_tp_kernel, sdp_fwd_lat_kernel, sdp_fwd_x_kernel:
        // full strips, no lengths) and with it (their _c twins: per-pair lengths or N not a multiple of 64; sdp_api.hip picks).  The
        // general-pitch and parts builds always carry it, and apply it where something foreign can be met: per-pair lengths and
        // partial strips.  (The groups of the general-pitch and latency builds straddle the ends of
        // every row, but into the pair's OWN plane -- the neighbouring rows of the same problem, the same on every run and in every
        // batch: the plane's first row starts a group exactly, r mod 4 = 0, and what follows its last row is past the buffer
        // descriptor's range and reads as zero.  Cleaning the first and last strip as well cost the latency build 3.8 % at
        // 64 x 512^2 -- those strips' ramps are the pair's critical path -- for nothing.)  No flag of it lives across the chunk loop.
        auto need_clean = [&]() {
            if constexpr (!CLEAN) return false;
            else return rows < 64 || p.lens != nullptr;
        };
        auto load_block_i = [&](int bb, auto plain_tag, int i, auto rset_tag) {  // instruction i of block set bb -> registers (set RS)
            constexpr bool plain = decltype(plain_tag)::value;
            constexpr int RS = decltype(rset_tag)::value;   // (a compile-time flag: with a run-time one every load sat behind its own branch)
            if constexpr (T::SIN > 0) {
                const int ubase = (i0 * ld + bb * K) * 4;
                // non-plain: a group that lies entirely left or right of the row would be a real fetch (of the
                // neighbouring row's data, 19 % of the input traffic at M = 512): send it out of range instead.
                // Groups that straddle column 0 or M are loaded (the cells outside are never used).
                const int col = li_col[i] + bb * K;
                // plain: the per-lane part must be a non-negative offset on its own (the hardware range-checks it before
                // the scalar part is added).  With blocks moved left (li_unaligned) it can be as low as -(2K + K - 1)
                // floats for the first rows, so 3K floats travel from the scalar part to the per-lane part (plain block
                // sets then have bb >= QMAX + 1 = 3, i.e. a scalar part of at least 3K floats).
                const int bias = li_unaligned ? 3 * K * 4 : 0;
                // (forward sweep: rows below a partial strip are not fetched)
                const bool rok = !CLEAN || (LINES ? ((i & 3) + 32 * (i >> 2) + 4 * r4_l) : (i * RPL + r4_l)) < rows;
                const unsigned off = plain ? li_voff[i] + (unsigned)bias : ((col > -4 && col < m && rok) ? li_voff[i] + (unsigned)ubase : OOB);
#pragma unroll
                for (int q = 0; q < T::SIN; ++q) {
                    if constexpr (ABL_NOLOAD) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) rs[RS][q][4 * i + j] = __uint_as_float((off + ubase + j) & 0x3fffffu) * 1e30f;
                    } else {
                        // each dword is range-checked on its own (tools/ubench/bufx4.hip), and only dword
                        // alignment is needed, so M need not be a multiple of 4
                        const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_in[q], off, plain ? ubase - bias : 0, LINES ? AUX_LINES_LOAD : AUX_IN_LOAD);
                        const unsigned v0 = v[0], v1 = v[1], v2 = v[2], v3 = v[3];
                        rs[RS][q][4 * i] = __uint_as_float(v0);
                        rs[RS][q][4 * i + 1] = __uint_as_float(v1);
                        rs[RS][q][4 * i + 2] = __uint_as_float(v2);
                        rs[RS][q][4 * i + 3] = __uint_as_float(v3);
                    }
                }
            }
        };
        using rs0_t = std::integral_constant<int, 0>;
        auto load_block_s = [&](int bb, auto rset_tag) {  // whole block set at once
            if (block_plain(bb)) {
#pragma unroll
                for (int i = 0; i < NLD; ++i) load_block_i(bb, std::true_type{}, i, rset_tag);
            } else {
#pragma unroll
                for (int i = 0; i < NLD; ++i) load_block_i(bb, std::false_type{}, i, rset_tag);
            }
        };
        auto load_block = [&](int bb) { load_block_s(bb, rs0_t{}); };
        // (ZSKIP_A) every value of plane 0 of register set RS is +0 / -0 in every lane
        auto block_zero = [&](auto rset_tag) -> bool {
            constexpr int RS = decltype(rset_tag)::value;
            unsigned any = 0;
#pragma unroll
            for (int k = 0; k < K; ++k) any |= __float_as_uint(rs[RS][0][k]) << 1;   // (the sign does not matter: e = -0 multiplies like +0 here)
            return __builtin_amdgcn_ballot_w64(any != 0) == 0;
        };
        auto write_block_c = [&](int bb, auto rset_tag, auto clean_tag) {  // registers -> LDS ring
            constexpr int RS = decltype(rset_tag)::value;
            constexpr bool CLEAN = decltype(clean_tag)::value;   // elements outside the row's columns [0, m) become 0 (see need_clean)
            if constexpr (T::SIN > 0 && !ABL_NOLDS) {
                const int flip = (bb & 1) * K;
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    if constexpr (CLEAN) {
                        const int col = li_col[i] + bb * K;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const bool ok = (unsigned)(col + j) < (unsigned)m;
#pragma unroll
                            for (int q = 0; q < T::SIN; ++q) rs[RS][q][4 * i + j] = ok ? rs[RS][q][4 * i + j] : 0.f;
                        }
                    }
#pragma unroll
                    for (int q = 0; q < T::SIN; ++q) {
                        if constexpr (LINES && !GEN) {
                            // aligned planes and pitch: ring position of the group's first column = r mod 4 = i mod 4 (mod 4)
                            float *dst = lds_in + q * PLANE;
                            const float v0 = rs[RS][q][4 * i], v1 = rs[RS][q][4 * i + 1], v2 = rs[RS][q][4 * i + 2], v3 = rs[RS][q][4 * i + 3];
                            if ((i & 3) == 0) {
                                *reinterpret_cast<float4 *>(dst + (li_w[i][0] ^ flip)) = make_float4(v0, v1, v2, v3);
                            } else if ((i & 3) == 2) {
                                *reinterpret_cast<float2 *>(dst + (li_w[i][0] ^ flip)) = make_float2(v0, v1);
                                *reinterpret_cast<float2 *>(dst + (li_w[i][2] ^ flip)) = make_float2(v2, v3);
                            } else {
                                // (the middle pair as ds_write2_b32 -- two data registers of its own -- not ds_write_b64: elements 1, 2 of a
                                //  loaded dwordx4 are an ODD-aligned register pair, a 64-bit operand wants an even one, and the compiler makes
                                //  one by copying -- 16 moves per chunk, and, whenever register pressure tips it that way (round 6 saw it with
                                //  two unrelated changes), it places the copies right BEHIND the loads with a wait for them: the whole memory
                                //  latency exposed at the top of every chunk, forward sweep 171 -> 194 us)
                                dst[li_w[i][0] ^ flip] = v0;
                                asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" : : "v"((unsigned)(uintptr_t)(dst + (li_w[i][1] ^ flip))), "v"(v1), "v"(v2) : "memory");
                                dst[li_w[i][3] ^ flip] = v3;
                            }
                        } else if constexpr (LINES) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) lds_in[q * PLANE + (li_w[i][j] ^ flip)] = rs[RS][q][4 * i + j];
                        } else {
                            *reinterpret_cast<float4 *>(lds_in + q * PLANE + (li_w[i][0] ^ flip)) =
                                make_float4(rs[RS][q][4 * i], rs[RS][q][4 * i + 1], rs[RS][q][4 * i + 2], rs[RS][q][4 * i + 3]);
                        }
                    }
                }
            }
        };
        auto write_block_s = [&](int bb, auto rset_tag) {
            if constexpr (CLEAN) {
                // ... and only block sets that can hold a group STRADDLING an end of its row: groups wholly outside were not fetched
                // (zeros), groups wholly inside need nothing.  A set's groups of row r start at columns bb K - K q_r - delta_r (+ 4 cg):
                // within (bb - 3) K .. bb K for the aligned build (whose groups start on multiples of four: no straddle at column 0,
                // and none at m unless m mod 4 != 0), within (bb - 4) K .. bb K for the general-pitch builds, (bb - 5) K .. for the latency builds.
                constexpr int BACK = !LINES ? 5 : (GEN ? 4 : 3);   // (latency builds: q_r up to 4 at K = 16, and their groups start up to 3 columns further left)
                const bool edge_r = m > (bb - BACK) * K && m < (bb + 1) * K && (GEN || !LINES || (m & 3) != 0);
                const bool edge_l = (GEN || !LINES) && bb < BACK;
                if (!block_plain(bb) && need_clean() && (edge_l || edge_r)) {
                    write_block_c(bb, rset_tag, std::true_type{});
                    return;
                }
            }
            write_block_c(bb, rset_tag, std::false_type{});
        };
        auto write_block = [&](int bb) { write_block_s(bb, rs0_t{}); };

        const int c_first = REV ? nchunks - 1 : 0;
        const int dir = REV ? -1 : 1;

        // ---- prologue ----
        if constexpr (T::QIN == Q_EXACT) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if constexpr (ABL_NOLOAD) rqx2[0][k] = rqx2[1][k] = make_float2(0.25f + 1e-3f * k, 0.5f - 1e-3f * lane);
                else if constexpr (!LAZY) rqx2[0][k] = load_f2(rs_qx, c_first * K, k);   // (LAZY: below)
            }
        }
        if constexpr (T::QIN == Q_PACKED) {
#pragma unroll
            for (int jj = 0; jj < QROWS; ++jj) {
                if constexpr (ABL_NOLOAD) rq2[0][jj] = rq2[1][jj] = (u32x4q){0x20003000u + 64 * jj + lane, 0x20003000u, 0x20003000u, 0x20003000u};
                else if constexpr (!LAZY) load_q20(c_first * K, jj, std::integral_constant<int, 0>{});
            }
        }
        if constexpr (T::DIN) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if constexpr (ABL_NOLOAD) rdd2[0][k] = rdd2[1][k] = make_float2(0.25f + 1e-3f * k, 0.5f - 1e-3f * lane);
                else rdd2[0][k] = load_d(c_first * K, k);
            }
        }
        zring = 0;
        if constexpr (LAZY) {
            // the strip's first chunk: its carries are the initial ones; if they and everything that reaches it from outside are
            // known to be +0 it is a zero chunk and its rows are not fetched.  Not waited for: the strip below is usually a lag
            // behind at this point, the fetch overlaps that wait, and a fetch held back until the boundary values are there
            // would put its latency on every hand-off of the ramp (measured: +7 us at 256 x 512^2).
            const unsigned cbits = __float_as_uint(cy.fa) | __float_as_uint(cy.fb) | __float_as_uint(cy.fc);
            known_zero = __builtin_amdgcn_ballot_w64(cbits != 0) == 0 && boundary_zero(c_first * K, false);
            load_rows(c_first * K, std::integral_constant<int, 0>{}, !known_zero);
        }
        load_block(c_first);
        write_block(c_first);
        if constexpr (ZSKIP_A) za_b1 = block_zero(rs0_t{});
        load_block(c_first + 1);
        write_block(c_first + 1);
        if constexpr (ZSKIP_A) {
            za_b2 = block_zero(rs0_t{});
            known_zero = false;                     // (the first chunk's rows are fetched whatever it turns out to be)
            load_block_s(c_first - 1, rs0_t{});     // the staged blocks travel two chunks ahead: the first body finds this one in set 0
        }
        // Every load of the prologue has returned before the chunk loop is entered (vmcnt(0), expcnt / lgkmcnt untouched) -- said
        // with the builtin, which the compiler's own wait-count bookkeeping understands.  Without it the state records of the
        // FIRST chunk are "pending loads into v[0:31]" on the loop-entry path only; the compiler merges that with the back
        // edge, and every iteration then waited (`s_waitcnt vmcnt(1)` in front of the first reuse of v0 as a temporary) until
        // all but one of the loads it had just issued for the NEXT chunk were back: the prefetch distance was not a chunk but
        // the few hundred cycles up to that point, and the reverse sweeps ran at memory latency (round 4, cycle stamps:
        // 900-2200 cycles per chunk in a phase that issues eight LDS writes).  The price: one exposed latency per strip.
        if constexpr (REV || T::QIN == Q_EXACT) __builtin_amdgcn_s_waitcnt(0x0F70);

        // ---- forward sweep, exp domain: the K steps of a chunk are computed in blocks of WB = 16 ----
        // The chunk (K steps) stays the unit of the memory pipeline -- staged input blocks, state prefetch distance --
        // but everything the recurrence itself touches is per block: the block's inputs (2 x 16 values from the LDS
        // ring), the 16 boundary values of the strip above (waited for, read and converted per block), the 16 values
        // handed to the strip below (published with their frame word and the progress word per block, which also
        // shortens the lag between strips from 63+K to 63+16 steps).  Arrays of 16 instead of K entries keep the
        // sweep's working set in VGPRs (the K = 32 build had ~340 instructions per chunk that only moved values
        // between VGPRs and AGPRs).  A block first runs in the windowed form (frame shared by its 16 steps); if a
        // value leaves the safe range nothing was committed and the block is redone in the per-step-normalised form.
        // Both forms produce identical bits (same 2^theta, exact power-of-two rescaling), so results do not depend on
        // which form a block ran in, on K, or on the batch.
        constexpr bool FWD_SUB = PASS == PASS_FWD && !ABL_NOMATH;
        auto read_inputs = [&](int tb, float *d0, float *d1) {
            const int pr = (tb & (RING - 1)) + 4 * ring_pi(lane & 7);
#pragma unroll
            for (int g = 0; g < WB / 4; ++g) {
                const int idx = lane * PITCH + ((pr + 4 * g) & (RING - 1));
                const float4 v0 = *reinterpret_cast<const float4 *>(lds_in + idx);
                const float4 v1 = *reinterpret_cast<const float4 *>(lds_in + PLANE + idx);
                d0[4 * g] = v0.x, d0[4 * g + 1] = v0.y, d0[4 * g + 2] = v0.z, d0[4 * g + 3] = v0.w;
                d1[4 * g] = v1.x, d1[4 * g + 1] = v1.y, d1[4 * g + 2] = v1.z, d1[4 * g + 3] = v1.w;
            }
        };
        // Boundary row of the block-wise forward sweep: the 8 bytes per column are kept as two PLANES -- mcap values (float)
        // followed by mcap exponents (int) -- instead of (value, exponent) pairs.  A block published in one frame (the
        // normal case) writes and reads its 16 values only: four 16-byte LDS accesses instead of sixteen 8-byte ones on
        // either side (a wave issues one LDS instruction per ~20 cycles, and the 64 of them per chunk that the boundary
        // cost were a tenth of the sweep's time); the exponents of such a block ARE its frame word.  Values that could not
        // be put into one frame carry their exponents in the second plane.
        const float *bv_in = reinterpret_cast<const float *>(bnd_in);
        const int *be_in = reinterpret_cast<const int *>(bnd_in) + p.mcap;
        float *bv_out = reinterpret_cast<float *>(bnd_out);
        int *be_out = reinterpret_cast<int *>(bnd_out) + p.mcap;
        // ---- bridge between parts (parts): a boundary that crosses workgroups travels through global memory as 8-byte
        // GRANULES {value, tag}: one naturally aligned 8-byte store per column (write-through, sc0 sc1), so a granule is
        // either wholly the memset pattern or wholly written, and neither side needs a flag or a fence (the tag -- the
        // exponent in the forward sweep, 0 in the fp32 backward sweep -- can never equal the pattern).  The PRODUCER
        // publishes into its LDS row as always and then copies what it published to the bridge row, one granule per lane,
        // one store instruction per block / chunk.  The CONSUMER replays the producer's publishing into its own LDS row
        // before it looks at the row -- values, frame word, progress word, exactly what a strip of its own workgroup would
        // have written -- so everything behind the LDS protocol is unchanged.  Granules are loaded one block / chunk ahead.
        const int xb_in_k = REV ? part_pos : part_pos - 1, xb_out_k = REV ? part_pos - 1 : part_pos;   // boundary k lies between part positions k, k + 1
        __amdgpu_buffer_rsrc_t rs_xi = make_rsrc(imported ? (const void *)(p.xb + ((size_t)b * (p.nparts_max - 1) + xb_in_k) * p.xb_row) : (const void *)p.vout,
                                                 imported ? (unsigned)p.xb_row * 8u : 0u);
        __amdgpu_buffer_rsrc_t rs_xo = make_rsrc(exported ? (const void *)(p.xb + ((size_t)b * (p.nparts_max - 1) + xb_out_k) * p.xb_row) : (const void *)p.vout,
                                                 exported ? (unsigned)p.xb_row * 8u : 0u);
        constexpr int XB_AUX = 17;   // sc0 sc1: stores write through to memory, loads are served from memory (no XCD's L2 in between)
        constexpr int XN = (PASS == PASS_FWD) ? WB : K;   // column granules per unit: a 16-step block (forward) / a chunk (reverse)
        // the forward sweep's blocks also have a FRAME word (the exponent all the block's columns were published in, or
        // FRAME_NONE), and which form the consumer's block takes -- its rounding -- depends on it: it travels as one more
        // granule {frame, 0} per block, behind the columns of the row, written and read by lane XN in the same instructions
        constexpr int XV = (PASS == PASS_FWD) ? XN + 1 : XN;   // lanes that hold a granule
        const int xb_frm0 = xb_frame_base(p.M);
        auto xb_off = [&](int unit) -> unsigned {
            if (lane < XN) return (unsigned)((unit * XN + lane) * 8);
            return (PASS == PASS_FWD && lane == XN) ? (unsigned)((xb_frm0 + unit) * 8) : OOB;
        };
        // granules are loaded TWO units ahead of their use, into two register sets that take turns (set = unit & 1): a
        // load that bypasses the L2s takes 1-2 us, a block of the forward sweep ~1.5 us
        unsigned xg_lo[2] = {0u, 0u}, xg_hi[2] = {XB_INVALID, XB_INVALID};   // lanes < XN hold one granule each
        int xg_unit[2] = {-(1 << 30), -(1 << 30)};                             // which unit a set holds
        int ximp = (PASS == PASS_FWD) ? 0 : (m - 1) / K;   // next unit to import: producer block (ascending) / producer chunk (descending)
        constexpr int XDIR = (PASS == PASS_FWD) ? 1 : -1;
        auto xb_issue_set = [&](auto set_tag, int unit) {
            constexpr int st = decltype(set_tag)::value;
            const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs_xi, unit >= 0 ? xb_off(unit) : OOB, 0, XB_AUX);
            const unsigned v0 = v[0], v1 = v[1];
            xg_lo[st] = v0, xg_hi[st] = v1, xg_unit[st] = unit;
        };
        auto xb_issue = [&](int unit) {
            if (unit & 1) xb_issue_set(std::integral_constant<int, 1>{}, unit);
            else xb_issue_set(std::integral_constant<int, 0>{}, unit);
        };
        // -> the unit's granules (value, tag) once every one of them is there (bounded spin, reported like a missed LDS
        // hand-off).  A consumer that has caught up with its producer finds its early loads still unwritten and has to ask
        // again -- a round trip to memory for EVERY unit from then on, since it keeps running right behind the producer.
        // So a consumer that had to wait lets the producer get XB_LAG units ahead before it carries on (once; if it
        // catches up again, again): after that the early loads hit.
        constexpr int XB_LAG = 4;
        const int x_last = (PASS == PASS_FWD) ? nchunks * (K / WB) - 1 : 0;   // the last unit the producer writes
        auto xb_wait = [&](int unit, int c, unsigned &glo, unsigned &ghi) {
            const int st = unit & 1;
            if ((st ? xg_unit[1] : xg_unit[0]) != unit) xb_issue(unit);
            glo = st ? xg_lo[1] : xg_lo[0], ghi = st ? xg_hi[1] : xg_hi[0];
            bool ok = __builtin_amdgcn_ballot_w64(lane < XV && ghi == XB_INVALID) == 0;
            if (!ok) {
                const int spin_cap = (SDP_EXP_BUILD && (p.dbg & 8)) ? (1 << 8) : (1 << 19);
                const int ahead = (PASS == PASS_FWD) ? (unit + XB_LAG < x_last ? unit + XB_LAG : x_last) : (unit - XB_LAG > 0 ? unit - XB_LAG : 0);
                for (int spin = 0; spin < spin_cap && !ok; ++spin) {   // the unit ahead is written after the unit itself
                    __builtin_amdgcn_s_sleep(4);
                    const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs_xi, xb_off(ahead), 0, XB_AUX);
                    const unsigned v1 = v[1];
                    ok = __builtin_amdgcn_ballot_w64(lane < XV && v1 == XB_INVALID) == 0;
                }
                xb_issue(unit);
                glo = st ? xg_lo[1] : xg_lo[0], ghi = st ? xg_hi[1] : xg_hi[0];
                ok = ok && __builtin_amdgcn_ballot_w64(lane < XV && ghi == XB_INVALID) == 0;
                if (!ok && p.status && lane == 0) {
                    if (__hip_atomic_fetch_add(p.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
                        p.status[1] = b, p.status[2] = s, p.status[3] = c | (PASS << 24);
                    }
                }
                xb_issue(unit + XDIR);   // (its early load predates the wait)
            }
            xb_issue(unit + 2 * XDIR);   // two units ahead, into the set this unit leaves
            return ok;
        };
        auto read_boundary = [&](int tb, float *bcf, int &fa, int &fb) {   // lane 0 at step tb+j needs column tb+j
            if (has_pred && tb < m) {
                if (tb + WB <= m) {
#pragma unroll
                    for (int g = 0; g < WB / 4; ++g) {   // (tb is a multiple of 16 and the rows are 16-byte aligned: ds_read_b128)
                        const float4 v = *reinterpret_cast<const float4 *>(__builtin_assume_aligned(bv_in + tb + 4 * g, 16));
                        bcf[4 * g] = v.x, bcf[4 * g + 1] = v.y, bcf[4 * g + 2] = v.z, bcf[4 * g + 3] = v.w;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < WB; ++j) bcf[j] = (tb + j < m) ? bv_in[tb + j] : EXP_ONE_A;
                }
                fa = frm_in[(tb + 63) / WB];                    // block that produced column tb
                fb = tb + 1 < m ? frm_in[(tb + 64) / WB] : fa;  // ... columns tb+1 .. tb+WB-1
            } else {
#pragma unroll
                for (int j = 0; j < WB; ++j) bcf[j] = EXP_ONE_A;
            }
        };
        // (value, exponent) pairs for the per-step form: the exponent of a column is its block's frame word, or -- for a
        // block that was not published in one frame -- what the exponent plane holds
        auto boundary_pairs = [&](int tb, const float *bcf, int fa, int fb, u64 *bcv) {
            if (has_pred && tb < m) {
#pragma unroll
                for (int j = 0; j < WB; ++j) {
                    const int f = j == 0 ? fa : fb;
                    int e = f;
                    if (f == FRAME_NONE) e = (tb + j < m) ? be_in[tb + j] : EXP_ONE_E;
                    bcv[j] = (tb + j < m) ? pack2(__float_as_uint(bcf[j]), (unsigned)e) : edge_zero<KIND>();
                }
            } else {
#pragma unroll
                for (int j = 0; j < WB; ++j) bcv[j] = edge_zero<KIND>();
            }
        };
        auto fwd_blocks = [&](int c, int t0) {
            if constexpr (FWD_SUB) {
                int thr = lane + (sw ? 1 : 0);  // EDGE: the lane's cell is live at step t iff t >= thr
                if (sw && i0 + lane == 0) thr = 0x7fffffff;  // padded row 1 of Smith-Waterman never is
                auto one_block = [&](auto sb_tag) {
                    constexpr int sb = decltype(sb_tag)::value;
                    const int tb = t0 + sb * WB;
                    const bool blk_interior = plain_strip && tb >= 63 && tb + WB < m;
                    // experiments build, sdp_set_trace: shader-cycle stamps of this block -- [pair / 64][wave][strip round][block][4]
                    auto stamp = [&](int k) {
                        if constexpr (SDP_EXP_BUILD != 0) {
                            if (p.trace && !(p.dbg & 1024) && (b & 63) == 0 && b < 256 && lane == 0 && (parts ? part < 4 : sidx / W < 2) && tb / WB < 40)
                                p.trace[((((b >> 6) * 4 + wave) * 4 + (parts ? part : sidx / W)) * 40 + tb / WB) * 8 + k] = __builtin_readcyclecounter();   // (k < 4 used)
                        }
                    };
                    stamp(0);
                    // ---- boundary values of this block: lane 0 at step tb+j needs column tb+j ----
                    // (the windowed form turns the registers into its `up` values in place, and the rare fallback simply
                    // reads the row again -- the strip above cannot overwrite these columns before this strip has
                    // produced its own)
                    const bool use_pred = has_pred && tb < m;
                    // One LDS round trip per block instead of four: the progress word of the strip above is read FIRST,
                    // the boundary values, their frame words and this block's inputs are read right behind it, and there
                    // is a single wait.  LDS executes a wave's reads in order, so if the progress word already covers the
                    // block, the values read behind it are the published ones; otherwise (only while the pipeline fills)
                    // the wave spins as before and reads them again.
                    float bcf0[WB];
                    float in0[WB], in1[WB];
                    int fa0 = 0, fb0 = 0, prog_seen = 0;
                    const int need = tb + WB < m ? tb + WB : m;
                    if (use_pred && imported) {
                        // replay the producer blocks this block depends on: block j published columns 16 j - 63 .. 16 j - 48 and
                        // left the progress word at 16 j - 47 (clipped to [0, m])
                        const int jneed = (need + 47 + 15) / 16;
                        float *bv_w = reinterpret_cast<float *>(bnd + (size_t)pslot * p.mcap);
                        int *be_w = reinterpret_cast<int *>(bv_w) + p.mcap;
                        int *frm_w = frm + pslot * FRAME_CAP;
                        while (ximp <= jneed) {
                            unsigned xg_lo = 0, xg_hi = 0;
                            xb_wait(ximp, c, xg_lo, xg_hi);
                            const int col = 16 * ximp - 63 + lane;
                            const bool colok = lane < WB && col >= 0 && col < m;
                            if (colok) bv_w[col] = __uint_as_float(xg_lo), be_w[col] = (int)xg_hi;
                            const int frame = __builtin_amdgcn_readlane((int)xg_lo, XN);   // the block's frame word, as the producer wrote it
                            const int hi_ = 16 * ximp - 47, done_ = hi_ < 0 ? 0 : (hi_ > m ? m : hi_);
                            if (lane == 0) {
                                frm_w[ximp] = frame;
                                lds_store_i32(prog + 4 * pword, done_);   // behind the values: LDS executes a wave's instructions in order
                            }
                            ++ximp;
                        }
                    }
                    if (use_pred) {
                        if constexpr (!ABL_NOSYNC) prog_seen = lds_issue_i32(prog + 4 * pword);
                    }
                    read_boundary(tb, bcf0, fa0, fb0);
                    read_inputs(tb, in0, in1);
                    if (use_pred) {
                        bool ready = ABL_NOSYNC;
                        if constexpr (!ABL_NOSYNC) {
                            lds_wait(prog_seen);
                            ready = __builtin_amdgcn_readfirstlane(prog_seen) >= pbase + need;
                        }
                        if (!ready) {
                            // bounded spin: a missed hand-off must never hang the device (~0.2 s at the cap).  If it ever
                            // gives up the results of this pair are wrong, so the wave says so in the status words the host
                            // checks on its next call (sdp_api.hip: SDP_E_HANDOFF) -- it does not carry on silently.
                            const int spin_cap = (SDP_EXP_BUILD && (p.dbg & 8)) ? (1 << 10) : (1 << 21);
                            for (int spin = 0; spin < spin_cap; ++spin) {
                                if (__builtin_amdgcn_readfirstlane(lds_load_i32(prog + 4 * pword)) >= pbase + need) {
                                    ready = true;
                                    break;
                                }
                                __builtin_amdgcn_s_sleep(1);
                            }
                            if (!ready && p.status && lane == 0) {
                                if (__hip_atomic_fetch_add(p.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
                                    p.status[1] = b, p.status[2] = s, p.status[3] = c | (PASS << 24);
                                }
                            }
                            read_boundary(tb, bcf0, fa0, fb0);
                        }
                    }
                    u64 hist[WB];
                    int frame_pub = FRAME_NONE;
                    stamp(1);

                    // ---- parts: the block's 16 columns (c_lo ..) to the bridge row as well, one granule per lane (columns outside
                    // the matrix: a valid dummy -- the consumer waits for whole blocks); frame_r: the publishing lane's frame ----
                    auto export_block = [&](int c_lo, int frame_r) {
                        const int col = c_lo + lane;
                        const bool colok = lane < WB && col >= 0 && col < m;
                        unsigned gv = __float_as_uint(EXP_ONE_A), ge = (unsigned)EXP_ONE_E;
                        const int fpub = __builtin_amdgcn_readlane(frame_r, PUB_LANE);   // (every lane has a frame of its own: the publishing lane's)
                        if (colok) {
                            gv = __float_as_uint(bv_out[col]);
                            ge = fpub != FRAME_NONE ? (unsigned)fpub : (unsigned)be_out[col];
                        }
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        if (!(SDP_EXP_BUILD && (p.dbg & 8)))
                            __builtin_amdgcn_raw_buffer_store_b64(lane == WB ? (u32x2){(unsigned)fpub, 0u} : (u32x2){gv, ge}, rs_xo,
                                                                  lane < WB ? (unsigned)((tb + lane) * 8) : (lane == WB ? (unsigned)((xb_frm0 + tb / WB) * 8) : OOB), 0, XB_AUX);
                    };
                    // ---- publish the block: 16 boundary values, their frame word, then the progress word.  Called at the end of
                    // whichever form computed the block -- NOT once behind both: with the 16 values (32 registers) flowing
                    // from two code paths into one publishing site the compiler shuffled them into common registers
                    // with 45-70 moves per block (a seventh of the block's instructions) ----
                    auto publish = [&]() {
                        stamp(2);
                    // ---- publish the block: 16 boundary values, their frame word, then the progress word ----
                        if (has_succ) {
                            const int c_lo = tb - 63;  // lane 63 produced column tb+j-63 at step j
                            if (lane == PUB_LANE) {
                                // values always; exponents only where the block is not in one frame (c_lo = tb - 63 is 1 mod 4:
                                // the 16-byte stores are not aligned, which LDS accepts -- one lane, off the critical path)
                                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                                // (written as asm: for a 4-byte aligned vector store the compiler emits dword stores)
                                auto st128 = [&](const void *dst, u32x4 v) {
                                    asm volatile("ds_write_b128 %0, %1" : : "v"((unsigned)(uintptr_t)dst), "v"(v) : "memory");
                                };
                                if (c_lo >= 0 && c_lo + WB <= m) {
#pragma unroll
                                    for (int g = 0; g < WB / 4; ++g)
                                        st128(bv_out + c_lo + 4 * g, (u32x4){lo32(hist[4 * g]), lo32(hist[4 * g + 1]), lo32(hist[4 * g + 2]), lo32(hist[4 * g + 3])});
                                    if (frame_pub == FRAME_NONE) {
#pragma unroll
                                        for (int g = 0; g < WB / 4; ++g)
                                            st128(be_out + c_lo + 4 * g, (u32x4){hi32(hist[4 * g]), hi32(hist[4 * g + 1]), hi32(hist[4 * g + 2]), hi32(hist[4 * g + 3])});
                                    }
                                } else {
#pragma unroll
                                    for (int j = 0; j < WB; ++j) {
                                        const int col = c_lo + j;
                                        if (col >= 0 && col < m) bv_out[col] = __uint_as_float(lo32(hist[j])), be_out[col] = (int)hi32(hist[j]);
                                    }
                                }
                                frm_out[tb / WB] = frame_pub;
                            }
                            const int hi = tb + WB - 63;
                            const int pub_done = hi < 0 ? 0 : (hi > m ? m : hi);
                            // LDS executes a wave's DS instructions in order, so the data written above is visible to any
                            // wave that observes this word (the asm statements also stop compiler reordering)
                            if (lane == PUB_LANE && !(SDP_EXP_BUILD && (p.dbg & 8))) lds_store_i32(prog + 4 * oword, obase + pub_done);
                            if (exported) export_block(c_lo, frame_pub);
                        }
                    };

                    // ---- windowed form; returns 1 = done, 0 = not applicable here, -1 = out of range ----
                    auto wf_block = [&](auto edge_tag, auto pred_tag) -> int {
                        constexpr bool EDGE = decltype(edge_tag)::value;
                        constexpr bool use_pred = decltype(pred_tag)::value;  // (shadows the run-time flag: one body per case)
                        int fa = 0, fb = 0;
                        if (use_pred) {
                            fa = __builtin_amdgcn_readfirstlane(fa0);
                            fb = __builtin_amdgcn_readfirstlane(fb0);
                            if (fa == FRAME_NONE || fb == FRAME_NONE) return 0;
                        }
                        int R = cy.xe + WF_BIAS;
                        if (use_pred && lane == 0) R = fb;
                        if constexpr (EDGE) {
                            const int nstarted = tb - (sw ? 1 : 0);  // lanes below this were live at step tb - 1
                            if (nstarted < 64) {  // the others take the frame of the last started lane (or of the boundary)
                                const int rref = nstarted > 0 ? __builtin_amdgcn_readlane(R, nstarted - 1) : (use_pred ? fb : EXP_ONE_E + WF_BIAS);
                                R = lane < nstarted ? R : rref;
                            }
                        }
                        float x = __builtin_amdgcn_ldexpf(cy.xa, cy.xe - R);
                        float d = __builtin_amdgcn_ldexpf(cy.da, cy.de - R);
                        const float xz = __builtin_amdgcn_ldexpf(EXP_ONE_A, EXP_ONE_E - R);  // V = 0 in this frame
                        const int Rn = dpp_i32<DPP_IN>(R, R);
                        // The neighbour's value enters this lane's frame as ldexp(ua, Rn - R) -- NOT as ua * 2^(Rn - R): the factor
                        // alone underflows to zero for Rn - R < -149 while the product need not (ua reaches 2^110 in its own frame,
                        // and lane 0's frame is the PRODUCER's, which may lie far from its own exponent).  Found by the 1200-case
                        // soak of round 5 (tools/cases/fuzz2_1172.npz: SW, theta x 8, 71 x 81, one pair of 133, plane starting 3
                        // floats off a 16-byte boundary: lane 1's `up` weight at (65, 79) came out 0 instead of 0.975, Vt off by
                        // 0.033, E by 5e-3; present since the windowed form exists).  Same bits wherever nothing underflowed.
                        const int dR = Rn - R;
                        unsigned mx = max(__float_as_uint(x), __float_as_uint(d)), mn = 0x3f800000u, mc = 0;
                        if (!EDGE || tb > thr) mn = __float_as_uint(x);  // a value that is not live yet may be arbitrarily small
                        float bf[WB];  // lane 0's `up` values in its frame
                        if (use_pred) {
                            bf[0] = __builtin_amdgcn_ldexpf(bcf0[0], fa - R);
#pragma unroll
                            for (int j = 1; j < WB; ++j) bf[j] = bcf0[j];
                        } else {
#pragma unroll
                            for (int j = 0; j < WB; ++j) bf[j] = xz;
                        }
                        // 2^theta, 2^A of the block: the log2(e) scalings as packed multiplies (two values per instruction)
                        // Range of the factors, tested on the exponents: |theta log2e| <= 12 (two-sided) and A log2e <= 12.
                        // With 2^-12 <= 2^theta <= 2^12 the per-step test on the lane's own value x is enough: the sum it
                        // was made from is x / 2^theta, i.e. within [2^-112, 2^122] -- normal, with a normal reciprocal --
                        // whatever the neighbour's scaled values were (an overflow or NaN anywhere ends up in x).
                        float ctv[WB], cav[WB];
                        float mcf = 0.f;
#pragma unroll
                        for (int j = 0; j < WB; j += 2) {
                            const f32x2 tt = (f32x2){in0[j], in0[j + 1]} * (f32x2){1.44269504088896340736f, 1.44269504088896340736f};
                            const f32x2 ta = (f32x2){in1[j], in1[j + 1]} * (f32x2){1.44269504088896340736f, 1.44269504088896340736f};
                            ctv[j] = __builtin_amdgcn_exp2f(tt[0]), ctv[j + 1] = __builtin_amdgcn_exp2f(tt[1]);
                            cav[j] = __builtin_amdgcn_exp2f(ta[0]), cav[j + 1] = __builtin_amdgcn_exp2f(ta[1]);
                            if constexpr (QX) {
                                const f32x2 c = exp2_residual((f32x2){in0[j], in0[j + 1]}, tt);
                                const f32x2 e = __builtin_elementwise_fma((f32x2){ctv[j], ctv[j + 1]}, c, (f32x2){ctv[j], ctv[j + 1]});
                                ctv[j] = e[0], ctv[j + 1] = e[1];
                            }
                            mcf = __builtin_fmaxf(__builtin_fmaxf(mcf, __builtin_fabsf(tt[0])), __builtin_fabsf(tt[1]));
                            mcf = __builtin_fmaxf(__builtin_fmaxf(mcf, ta[0]), ta[1]);
                        }
                        // (a NaN in theta or A is ignored by the maxima above but turns x into NaN, whose bit pattern
                        // fails the upper test below)
                        mc = mcf <= 12.0f ? 0u : 0xffffffffu;
#pragma unroll
                        for (int j = 0; j < WB; ++j) {
                            const float ct = ctv[j], ca = cav[j];
                            const float ua = __uint_as_float(dpp_i32<DPP_IN>(__float_as_int(bf[j]), __float_as_int(x)));
                            const float u = __builtin_amdgcn_ldexpf(ua, dR);
                            const float ssum = __builtin_fmaf(ca, u + x, d);
                            const float rinv = __builtin_amdgcn_rcpf(ssum);
                            const float tq = ca * rinv;
                            if constexpr (QX) {
                                float2 qq = make_float2(tq * u, tq * x);
                                q_sharpen(qq.x, qq.y, d * rinv);
                                if constexpr (ABL_NOSTORE) { keep(qq.x); keep(qq.y); }
                                else store_state(tb, j, qq);
                            } else {
                                // both weights with one packed multiply, their biased fields with one packed fma
                                const f32x2 w = (f32x2){u, x} * (f32x2){tq, tq};
                                const f32x2 f = __builtin_elementwise_fma(w, (f32x2){QF_SCALE, QF_SCALE}, (f32x2){QF_BASE, QF_BASE});
                                if constexpr (ABL_NOSTORE) { float fx = f[0], fy = f[1]; keep(fx); keep(fy); }
                                else store_state_bits(tb, j, __float_as_uint(f[0]), __float_as_uint(f[1]));
                            }
                            d = u;
                            x = ct * ssum;
                            if constexpr (EDGE) {
                                const bool live = tb + j >= thr;
                                x = live ? x : xz;
                                mn = min(mn, live ? __float_as_uint(x) : 0x3f800000u);
                            } else {
                                mn = min(mn, __float_as_uint(x));
                            }
                            mx = max(mx, __float_as_uint(x));
                            hist[j] = pack2(__float_as_uint(x), (unsigned)R);
                        }
                        if (__builtin_amdgcn_ballot_w64(mx > WF_HI || mn < WF_LO || mc > WF_FMAX) != 0) return -1;  // carry untouched
                        cy.xa = __builtin_amdgcn_frexp_mantf(x);
                        cy.xe = R + __builtin_amdgcn_frexp_expf(x);
                        cy.da = __builtin_amdgcn_frexp_mantf(d);
                        cy.de = R + __builtin_amdgcn_frexp_expf(d);
                        if constexpr (EDGE) {
                            if (tb + WB - 1 < thr) cy.xa = EXP_ONE_A, cy.xe = EXP_ONE_E;  // still waiting: exactly V = 0
                            const int tf = m - 1 + rows - 1;  // step at which the last strip meets the terminal cell
                            if (s == nstrips - 1 && tf >= tb && tf < tb + WB) {
                                const int ktf = t_final - tb;  // only that lane has t_final >= 0
#pragma unroll
                                for (int j = 0; j < WB; ++j) vt_keep = (j == ktf) ? hist[j] : vt_keep;
                                // a terminal cell on the Smith-Waterman border is not live: V = 0 exactly (its value in
                                // the frame may have underflowed to 0, which would read as -inf)
                                if (t_final >= 0 && t_final < thr) vt_keep = edge_zero<KIND>();
                            }
                        }
                        frame_pub = R;
                        publish();
                        return 1;
                    };

                    // ---- per-step-normalised form (the verified fallback; cannot overflow for any finite input) ----
                    auto norm_block = [&](auto edge_tag) {
                        constexpr bool EDGE = decltype(edge_tag)::value;
                        u64 bcv[WB];
                        boundary_pairs(tb, bcf0, fa0, fb0, bcv);
#pragma unroll
                        for (int j = 0; j < WB; ++j) {
                            const int t = tb + j;
                            const int col = t - lane;
                            const bool dead = EDGE && sw && (col == 0 || (i0 + lane) == 0);  // SW: padded row 1 / col 1
                            // theta = (kt + ft) ln2, A = (ka + fa) ln2 with integer kt, ka; clamped to +-2^20 bits so that
                            // A = -inf (a forbidden gap) behaves like the reference's exp(-inf) = 0 instead of inf - inf
                            const float tt = __builtin_amdgcn_fmed3f(in0[j] * 1.44269504088896340736f, -1048576.f, 1048576.f);
                            const float ta = __builtin_amdgcn_fmed3f(in1[j] * 1.44269504088896340736f, -1048576.f, 1048576.f);
                            // Moderate exponents take mantissa and exponent of the SAME 2^tt the windowed form multiplies
                            // with, so that both forms produce identical bits
                            float et2 = __builtin_amdgcn_exp2f(tt);
                            const float ea2 = __builtin_amdgcn_exp2f(ta);
                            if constexpr (QX) {   // the same corrected 2^theta as the windowed form, bit for bit
                                const f32x2 c = exp2_residual((f32x2){in0[j], in0[j]}, (f32x2){tt, tt});
                                const f32x2 e = __builtin_elementwise_fma((f32x2){et2, et2}, c, (f32x2){et2, et2});
                                et2 = e[0];
                            }
                            const float kt = __builtin_floorf(tt), ka = __builtin_floorf(ta);
                            // The product theta * log2(e), rounded to fp32, is off by up to |tt| 2^-24 bits -- 9e-6 at
                            // |theta| = 100 -- and the weights of the three cells that read this V inherit it; over a few
                            // hundred soft cells that reaches 1e-4 in Vtd (and 4e-5 in E).  Beyond the windowed form's
                            // range (|tt| > 12; below it the two forms must agree bit for bit) the fraction is formed
                            // from the exact product: log2(e) as hi + lo, the integer part taken out inside the fma.
                            // -inf (a forbidden gap) ends at the clamp: 2^0 with the exponent -2^20.
                            constexpr float L_HI = 1.44269502162933349609375f, L_LO = 1.92596299112661746e-8f;
                            const float ft = __builtin_fmaf(in0[j], L_LO, __builtin_fmaf(in0[j], L_HI, -kt));
                            const float fa2 = __builtin_fmaf(in1[j], L_LO, __builtin_fmaf(in1[j], L_HI, -ka));
                            const float st = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(ft, -0.5f, 1.5f));
                            const float sa = __builtin_amdgcn_exp2f(__builtin_amdgcn_fmed3f(fa2, -0.5f, 1.5f));
                            const bool mt = __builtin_fabsf(tt) <= 12.f, ma = __builtin_fabsf(ta) <= 12.f;
                            const float ct = mt ? __builtin_amdgcn_frexp_mantf(et2) : st;
                            const int kti = mt ? __builtin_amdgcn_frexp_expf(et2) : (int)kt;
                            const float ca = ma ? __builtin_amdgcn_frexp_mantf(ea2) : sa;
                            const int kai = ma ? __builtin_amdgcn_frexp_expf(ea2) : (int)ka;
                            const float ua = __uint_as_float(dpp_i32<DPP_IN>((int)lo32(bcv[j]), __float_as_int(cy.xa)));
                            const int ue = dpp_i32<DPP_IN>((int)hi32(bcv[j]), cy.xe);
                            const int ex = ue + kai, ey = cy.xe + kai, ed = cy.de;
                            const int er = max(max(ex, ey), ed);
                            const float u = __builtin_amdgcn_ldexpf(ua, ex - er);
                            const float l = __builtin_amdgcn_ldexpf(cy.xa, ey - er);
                            const float d = __builtin_amdgcn_ldexpf(cy.da, ed - er);
                            const float ssum = __builtin_fmaf(ca, u + l, d);  // the operand with the largest exponent is unshifted
                            // (a cell nothing reaches -- forbidden gaps on every way in, the diagonal predecessor unreachable itself -- has ssum = 0:
                            //  its weights are then 0, 0 and, by the sharpening, 1 for the diagonal, as the reference's finite -1e10 borders give them;
                            //  1 / 0 made them NaN in the exact state, which the first-order sweeps never noticed (E is 0 there) and the adjoint sweeps
                            //  spread over the whole pair: Vtd = NaN in 3 of 4000 cases of round 5's last soak, thin problems with 20 % forbidden gaps)
                            const float rinv = ssum >= 1.1754944e-38f ? __builtin_amdgcn_rcpf(ssum) : 0.f;
                            const float tq = ca * rinv;
                            {
                                float2 qq = make_float2(tq * u, tq * l);
                                // every build sharpens here: a block lands in this form when its scores are steep, and
                                // steep scores are where paths saturate -- a packed weight left at 1 - 2^-23 instead of 1
                                // loses 1.7e-8 of E per step on average (7e-5 over the 4096 steps of a 2048 x 2048
                                // problem).  Which blocks run this form does not depend on the build (wf_skip counts
                                // blocks, not chunks), so results stay bit-identical across wave counts.
                                q_sharpen(qq.x, qq.y, d * rinv);
                                if constexpr (ABL_NOSTORE) { keep(qq.x); keep(qq.y); }
                                else if constexpr (QX) store_state(tb, j, qq);
                                else store_state_bits(tb, j, __float_as_uint(__builtin_fmaf(qq.x, QF_SCALE, QF_BASE)),
                                                      __float_as_uint(__builtin_fmaf(qq.y, QF_SCALE, QF_BASE)));
                            }
                            const float an = ct * ssum;
                            float na = __builtin_amdgcn_frexp_mantf(an);
                            int ne = er + kti + __builtin_amdgcn_frexp_expf(an);
                            cy.da = ua;
                            cy.de = ue;
                            if constexpr (EDGE) {
                                const bool live = col >= 0 && !dead;
                                na = live ? na : EXP_ONE_A;
                                ne = live ? ne : EXP_ONE_E;
                            }
                            cy.xa = na;
                            cy.xe = ne;
                            hist[j] = pack2(__float_as_uint(na), (unsigned)ne);
                            if constexpr (EDGE) vt_keep = (t == t_final) ? hist[j] : vt_keep;
                        }
                        // publish in one frame whenever the 16 values fit (exact rescaling to the exponent of the last
                        // one), so that the strip below can use the windowed form; only the publishing lane matters
                        if (has_succ) {
                            const int R = (int)hi32(hist[WB - 1]);
                            float av[WB];
                            unsigned mx = 0, mn = ~0u;
#pragma unroll
                            for (int j = 0; j < WB; ++j) {
                                av[j] = __builtin_amdgcn_ldexpf(__uint_as_float(lo32(hist[j])), (int)hi32(hist[j]) - R);
                                mx = max(mx, __float_as_uint(av[j]));
                                mn = min(mn, __float_as_uint(av[j]));
                            }
                            const bool fits = mx <= WF_HI && mn >= WF_LO;
                            if ((__builtin_amdgcn_ballot_w64(fits) >> PUB_LANE) & 1ull) {
#pragma unroll
                                for (int j = 0; j < WB; ++j) hist[j] = pack2(__float_as_uint(av[j]), (unsigned)R);
                                frame_pub = R;
                            }
                        }
                        publish();
                    };

                    bool done = false;
                    // (experiments build: sdp_set_debug bits 16-23 = 1 + block, bits 24-30 = strip position: that block runs in the normalised form)
                    const bool dbg_norm = SDP_EXP_BUILD && ((p.dbg >> 16) & 0xff) == (unsigned)(tb / WB + 1) && ((p.dbg >> 24) & 0x7f) == (unsigned)sidx;
                    int dbg_code = wf_skip > 0 ? 8 : 0;
                    if (wf_skip == 0 && !dbg_norm) {
                        const int rc = use_pred ? (blk_interior ? wf_block(std::false_type{}, std::true_type{}) : wf_block(std::true_type{}, std::true_type{}))
                                                : (blk_interior ? wf_block(std::false_type{}, std::false_type{}) : wf_block(std::true_type{}, std::false_type{}));
                        done = rc > 0;
                        dbg_code = rc > 0 ? 1 : (rc < 0 ? 2 : 4);
                        if (rc < 0) wf_skip = 2;  // values move too fast for one frame per block here: try again two blocks later (the same in every build)
                    } else if (wf_skip > 0) {
                        --wf_skip;
                    }
                    if constexpr (SDP_EXP_BUILD != 0) {   // which form the block ran in: 1 windowed, 2 out of range, 4 not applicable (frames), 8 skipped after a failure, 0 forced
                        if (p.trace && !(p.dbg & 1024) && (b & 63) == 0 && b < 256 && lane == 0 && (parts ? part < 4 : sidx / W < 2) && tb / WB < 40)
                            p.trace[((((b >> 6) * 4 + wave) * 4 + (parts ? part : sidx / W)) * 40 + tb / WB) * 8 + 6] = 100 + dbg_code;
                    }
                    if (!done) {
                        if (blk_interior) norm_block(std::false_type{});
                        else norm_block(std::true_type{});
                    }

                    stamp(3);
                };
                one_block(std::integral_constant<int, 0>{});
                if constexpr (K / WB > 1) one_block(std::integral_constant<int, 1>{});
                static_assert(K / WB <= 2, "blocks per chunk");
            }
        };

        // ---- output flush after a chunk (reverse sweeps) ----
        // `active` = false: nothing to flush yet (first iteration) -- the same instructions run with every offset out of range.
        // The flush is branch-free on purpose, and so is its place in the loop: the compiler computes the vmcnt of a wait for
        // a state record (loaded one iteration earlier) from the memory instructions it is SURE were issued since, and a
        // conditional flush counted as zero stores -- every chunk then waited until its own, just issued, output stores had
        // been acknowledged by memory before it touched the first record (`s_waitcnt vmcnt(14)` where 30 were in flight:
        // a write round trip per chunk, the 35 us between the backward sweep on cache-served tensors and on real ones).
        // PIPE (fp32 backward sweep, aligned K = 32 builds: the kernels of the headline shape).  Round 4's chunk was five phases in
        // a row -- flush the previous chunk's outputs (16 LDS reads, a wait, 16 stores), wait for the boundary values (an LDS
        // round trip for the progress word, another for the values), request the next rows, 32 steps, publish (9 LDS writes) --
        // and a wave that has a SIMD to itself hides none of their latencies: of ~6100 cycles per chunk ~2500 were the
        // recurrence (cycle stamps, profiles/r04_bwd_trace.txt).  Here the chunk is one pipeline:
        //   * the LDS reads of the previous chunk's flush are issued at the top of the iteration, the progress word and the K
        //     boundary values right behind them (speculatively: LDS executes a wave's reads in order, so if the progress word
        //     covers the chunk the values read behind it are the published ones) -- ONE wait for all of it;
        //   * the flush's 16 stores leave one per two steps INSIDE the step loop, out of registers;
        //   * the boundary values this strip hands on are written four at a time as soon as their steps are done, and only
        //     the progress word is left for the end of the chunk.
        // The wait for the state rows (requested an iteration ago) sits at the top of the iteration and counts the stores a
        // pipelined chunk issued behind that request (tail_st).
        constexpr bool PIPE = FLUSH2 && LAZY && !NOPIPE;   // (NOPIPE: the twin build for short pairs, see sdp_bwd_kernel below)
        static_assert(!FLUSH2 || T::SIN == 0 || stage_out_pad(PASS) > 0, "FLUSH2 writes one float in front of lds_out: a staged input plane before it needs the pad");
        static_assert(!PIPE || KF == K, "the pipelined flush moves a chunk's own block");
        // the plain flush in two halves: LDS -> registers, registers -> memory (all 64 x 32 elements are real cells)
        auto flush_read = [&](int par, bool zero, float2 *vals) {
            const int thr_l = K - 1 - f2_rl - f2_el;   // (k2 & 7) < thr_l  <=>  sfull < K - 1
            if (zero) {
#pragma unroll
                for (int k2 = 0; k2 < K / 2; ++k2) vals[k2] = make_float2(0.f, 0.f);
            } else if (par) {
#pragma unroll
                for (int k2 = 0; k2 < K / 2; ++k2) {
                    const int idx = ((k2 & 7) + 32 * (k2 >> 3)) * PO + (k2 & 7) + f2_l + ((k2 & 7) < thr_l ? K : -K);
                    vals[k2] = *reinterpret_cast<const float2 *>(__builtin_assume_aligned(lds_out + idx, 8));
                }
            } else {
#pragma unroll
                for (int k2 = 0; k2 < K / 2; ++k2)
                    vals[k2] = *reinterpret_cast<const float2 *>(__builtin_assume_aligned(lds_out + f2_l + ((k2 & 7) + 32 * (k2 >> 3)) * PO + (k2 & 7), 8));
            }
        };
        auto flush_store1 = [&](int t0, int k2, float2 v, unsigned lane_off) {   // lane_off: f2_g * 4, or OOB = this store is a dummy
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const int c_r = (k2 & 7) + 32 * (k2 >> 3);
            if constexpr (ABL_NOSTORE) { keep(v.x); keep(v.y); }
            else __builtin_amdgcn_raw_buffer_store_b64((u32x2){__float_as_uint(v.x), __float_as_uint(v.y)}, rs_out, lane_off,
                                                       (i0 * ld + t0) * 4 + (c_r * ld - 32 * (k2 >> 3)) * 4, AUX_OUT_STORE);
        };
        // an all-zero plain flush: 64 rows x 128 bytes of +0 as eight dwordx4 stores (eight rows each), no LDS
        auto flush_zero8 = [&](int t0) {
            typedef unsigned u32x4z __attribute__((ext_vector_type(4)));
            const unsigned lane_off = (unsigned)(((lane >> 3) * ld + 4 * (lane & 7)) * 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (!ABL_NOSTORE)
                    __builtin_amdgcn_raw_buffer_store_b128((u32x4z){0u, 0u, 0u, 0u}, rs_out, lane_off, ((i0 + 8 * j) * ld + t0 - (j >= 4 ? K : 0)) * 4, AUX_OUT_STORE);
            }
        };
        auto flush_out = [&](int t0, int par, bool active, bool zero = false) {   // zero: both halves of the ring are known to hold +0
            if constexpr (T::SOUT > 0) {
                const int ubase = (i0 * ld + t0) * 4;
                if constexpr (FLUSH2) {
                    // all 64 x 32 elements are real cells (rows of a full strip, columns t0 - 32 .. t0 + 31 inside the matrix): no
                    // per-lane tests, the global offset is one per-lane base plus a scalar, the LDS index one per-lane base plus a
                    // constant -- no vector arithmetic at all when the chunk has parity 0, a compare-and-select per row class when 1
                    const bool flush_plain = active && rows == 64 && t0 >= KF && t0 + KF <= m;
                    if (flush_plain && !PIPE) {   // (PIPE: plain flushes are deferred -- flush_read / flush_store1 -- and never get here)
                        if (zero) flush_zero8(t0);
                        else {
                            float2 vals[KF / 2];
#pragma unroll
                            for (int k4 = 0; k4 < KF / 4; ++k4)
#pragma unroll
                                for (int j = 0; j < 2; ++j) {
                                    const int cc = 4 * (k4 & 3) + 32 * (k4 >> 2);
                                    const int sfull = 4 * (k4 & 3) + f4_s + 2 * j;
                                    const int idx = cc * PO + 4 * (k4 & 3) + f4_l + 2 * j + (par ? (sfull < KF - 1 ? KF : -KF) : 0);
                                    vals[2 * k4 + j] = *reinterpret_cast<const float2 *>(__builtin_assume_aligned(lds_out + idx, 8));
                                }
#pragma unroll
                            for (int k4 = 0; k4 < KF / 4; ++k4) {
                                typedef unsigned u32x4f __attribute__((ext_vector_type(4)));
                                const int cc = 4 * (k4 & 3) + 32 * (k4 >> 2);
                                const float2 v0 = vals[2 * k4], v1 = vals[2 * k4 + 1];
                                if constexpr (ABL_NOSTORE) { keep(vals[2 * k4].x); keep(vals[2 * k4 + 1].y); }
                                else __builtin_amdgcn_raw_buffer_store_b128((u32x4f){__float_as_uint(v0.x), __float_as_uint(v0.y), __float_as_uint(v1.x), __float_as_uint(v1.y)},
                                                                            rs_out, (unsigned)(f4_g * 4), ubase + (cc * ld - 32 * (k4 >> 2)) * 4, AUX_OUT_STORE);
                            }
                        }
                        if constexpr (LAZY && !PIPE) __builtin_amdgcn_s_waitcnt(0x0F78);   // vmcnt(8): everything older than these K / 4 stores
                    } else if ((m & 1) == 0) {   // (uniform)
                        float2 vals[KF / 2];
                        // (the per-lane constants pass through an opaque copy: the compiler otherwise hoists the 16 + 32 index /
                        //  offset calculations of these two rare paths out of the chunk loop into the strip's set-up -- ~250
                        //  instructions per strip and ~100 registers held across the whole sweep, round 5 ISA)
                        int f2_rl = f2_rl_c, f2_el = f2_el_c, f2_l = f2_l_c, f2_g = f2_g_c;
                        asm volatile("" : "+v"(f2_rl), "+v"(f2_el), "+v"(f2_l), "+v"(f2_g));
#pragma unroll
                        for (int k2 = 0; k2 < KF / 2; ++k2) {
                            const int sfull = (k2 & 7) + f2_rl + f2_el;
                            const int d1 = sfull < KF - 1 ? KF : -KF;   // parity 1: the other half of the ring (sfull = K - 1: position -1)
                            const int idx = ((k2 & 7) + 32 * (k2 >> 3)) * PO + (k2 & 7) + f2_l + (par ? d1 : 0);
                            vals[k2] = *reinterpret_cast<const float2 *>(__builtin_assume_aligned(lds_out + idx, 8));
                        }
#pragma unroll
                        for (int k2 = 0; k2 < KF / 2; ++k2) {
                            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                            const int c_r = (k2 & 7) + 32 * (k2 >> 3);
                            const int row = c_r + f2_rl;
                            const int col = t0 - 32 * (k2 >> 3) + f2_el;   // even: the pair (col, col + 1) is inside or outside together (m even)
                            const bool ok = active && (unsigned)col < (unsigned)m && (i0 + row) < n;
                            const unsigned off = ok ? (unsigned)((f2_g + c_r * ld - 32 * (k2 >> 3)) * 4 + ubase) : OOB;
                            if constexpr (ABL_NOSTORE) { keep(vals[k2].x); keep(vals[k2].y); }
                            else __builtin_amdgcn_raw_buffer_store_b64((u32x2){__float_as_uint(vals[k2].x), __float_as_uint(vals[k2].y)}, rs_out, off, 0, AUX_OUT_STORE);
                        }
                        if constexpr (LAZY && !PIPE) __builtin_amdgcn_s_waitcnt(0x4F70);   // vmcnt(16)
                    } else {   // an odd number of columns (per-pair lengths): a column at a time, indices formed on the spot
                        int r_l = r_l_c, s_l = s_l_c;
                        asm volatile("" : "+v"(r_l), "+v"(s_l));
#pragma unroll
                        for (int k = 0; k < KF; ++k) {
                            const int row = k * (64 / KF) + r_l, rho = row & (KF - 1), sfull = rho + s_l;
                            const int idx = row * PO + sfull + (par ? (sfull >= KF ? -KF : KF) : 0);
                            const float v = lds_out[idx];
                            const int col = t0 - (row - rho) + s_l;
                            const bool ok = active && (unsigned)col < (unsigned)m && (i0 + row) < n;
                            const unsigned off = ok ? (unsigned)((row * ld - (row - rho) + s_l) * 4 + ubase) : OOB;
                            if constexpr (ABL_NOSTORE) { unsigned vv = __float_as_uint(v) ^ off; keep(vv); }
                            else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_out, off, 0, AUX_OUT_STORE);
                        }
                        if constexpr (LAZY && !PIPE) __builtin_amdgcn_s_waitcnt(0x8F70);   // vmcnt(32)
                    }
                } else {
                    if constexpr (GEN) {
                        // General pitch, every element a real cell (a full strip, the rows' blocks -- which start up to 63 columns left of
                        // t0 -- inside the matrix): the blocks are K-float pieces of MEMORY lines, so four columns per lane leave as one
                        // aligned dwordx4 -- K / 4 stores per chunk instead of K masked dword stores, and none of the per-lane index
                        // arrays (round 6: the dword flush was 2700-2900 of the 6100-7900 cycles of a chunk of the general-pitch
                        // backward sweeps, with real memory and cache-served alike: tools/bwd_trace.py 0 64 1022 1020).  Row and block
                        // offset are formed on the spot: rho_r as in the set-up above.
                        // K = 32 only: at K = 16 (the 8-wave latency builds, BASELINE configs[2] with per-pair lengths) four dwordx4 stores of
                        // sixteen 64-byte row pieces each were no gain over sixteen dword stores of four (interleaved A/B: 523 -> 542 us with the
                        // zero fill, 300-330 either way without); K = 32, steady state: 64 x 1022 x 1020 backward 355.5 -> 328.3 us,
                        // 256 x 500 x 516 147.6 -> 142.7, 256 x 1022 x 1020 (the memory system's) 404.5 -> 409.8
                        if (GENFAST && active && rows == 64 && t0 >= 64 && t0 + K <= m) {
                            constexpr int LPRF = K / 4, RPIF = 64 / LPRF;
                            const int rlf = lane / LPRF, e4 = 4 * (lane % LPRF);
                            typedef unsigned u32x4g __attribute__((ext_vector_type(4)));
                            u32x4g v[K / 4];
                            unsigned voff[K / 4];
#pragma unroll
                            for (int j = 0; j < K / 4; ++j) {
                                const int r = j * RPIF + rlf;
                                const int rho = (r * (1 - ld) - fo_beta) & (K - 1);
                                const int s0 = rho + e4 + par * K;
                                const float *row = lds_out + r * PO;
#pragma unroll
                                for (int jj = 0; jj < 4; ++jj) v[j][jj] = __float_as_uint(row[(s0 + jj) & (2 * K - 1)]);
                                voff[j] = (unsigned)((r * ld - r + rho + e4) * 4 + ubase);
                            }
#pragma unroll
                            for (int j = 0; j < K / 4; ++j) {
                                if constexpr (ABL_NOSTORE) { unsigned vv = v[j][0] ^ v[j][3] ^ voff[j]; keep(vv); }
                                else __builtin_amdgcn_raw_buffer_store_b128(v[j], rs_out, voff[j], 0, AUX_OUT_STORE);
                            }
                            if constexpr (LAZY) __builtin_amdgcn_s_waitcnt(0x0F70 | ((K / 4) & 15));   // vmcnt(K / 4)
                            return;
                        }
                    }
                    float vals[K];
                    int g_rl = r_l, g_sl = s_l;   // (GENFAST: opaque copies, or the compiler hoists the index arithmetic out of the chunk loop again)
                    if constexpr (GENFAST) asm volatile("" : "+v"(g_rl), "+v"(g_sl));
                    auto g_rho = [&](int row) { return (row * (1 - ld) - fo_beta) & (K - 1); };
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        if constexpr (GENFAST) {
                            const int row = k * RPI + g_rl;
                            vals[k] = lds_out[row * PO + ((g_rho(row) + g_sl + par * K) & (2 * K - 1))];
                        } else {
                            vals[k] = lds_out[fo_off0[k] + par * fo_dk[k]];
                        }
                    }
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const int row = k * RPI + (GENFAST ? g_rl : r_l);
                        unsigned voff_k;
                        if constexpr (GENFAST) voff_k = (unsigned)((row * ld - (row - g_rho(row)) + g_sl) * 4);
                        else voff_k = fo_voff[k];
                        const int col = t0 + (int)(voff_k >> 2) - row * ld;   // t0 - D_row + s_l (voff >= 0: D_r <= r)
                        const bool ok = active && (unsigned)col < (unsigned)m && (i0 + row) < n;
                        const unsigned off = ok ? voff_k + (unsigned)ubase : OOB;
                        if constexpr (ABL_NOSTORE) {
                            unsigned vv = __float_as_uint(vals[k]) ^ off;
                            keep(vv);
                        } else {
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vals[k]), rs_out, off, 0, AUX_OUT_STORE);
                        }
                    }
                    if constexpr (LAZY) __builtin_amdgcn_s_waitcnt(0x0F70 | (K & 15) | ((K >> 4) << 14));   // vmcnt(K)
                }
            }
        };

        // Reverse sweeps flush the output blocks of a chunk at the top of the NEXT iteration (still before that chunk's
        // steps overwrite the ring half the flush reads), so that the flush code exists once: the loop runs one extra
        // iteration for the last chunk's flush, and one more for the tail flush at t0 = -K when the row pitch or the
        // plane is not aligned to K floats (rows whose blocks start to the right of chunk 0 still hold their first
        // columns in the ring; every element of that flush that exists was produced in chunk 0).
        const int nflush = T::SOUT > 0 ? (fo_need_tail ? 2 : 1) : 0;
        int pf_t0 = 0, pf_par = 0;
        // ROT: sweeps that prefetch state rows run the chunk loop unrolled by two, and the two instances use the two register
        // sets in opposite roles (P = set of the current chunk's rows, 1 - P = set the next chunk's rows are loaded into): no
        // row is ever moved, and a row is waited for where it is first used, a whole iteration and more behind its load.
        // (With one body and a move "next -> current" at its end the wait sat at the move.)
        constexpr bool ROT = TOPLOAD || TOPLOAD_X;
        auto chunk_body = [&](auto p_tag, const int ci) {
            constexpr int P = decltype(p_tag)::value;
            using cur_t = std::integral_constant<int, P>;
            using nxt_t = std::integral_constant<int, ROT ? 1 - P : P>;
            // experiments build, sdp_set_trace, backward sweep: per chunk (slot = position in processing order) cycle stamps
            // 0 top, 1 previous chunk's outputs flushed, 2 boundary values there, 3 steps done, 4 published
            // (debug bit 1024: the backward sweep's; 8192 / 16384: the adjoint backward / forward sweep's, tools/adj_trace.py)
            auto stamp_rev = [&](int k) {
                if constexpr (SDP_EXP_BUILD != 0 && (PASS == PASS_BWD || PASS == PASS_ABWD || PASS == PASS_AFWD)) {
                    if (p.trace && (p.dbg & (PASS == PASS_BWD ? 1024 : (PASS == PASS_ABWD ? 8192 : 16384))) && (b & 63) == 0 && b < 256 && lane == 0 && (parts ? part < 4 : sidx / W < 2) && ci < 40)
                        p.trace[((((b >> 6) * 4 + wave) * 4 + (parts ? part : sidx / W)) * 40 + ci) * 8 + k] = __builtin_readcyclecounter();
                }
            };
            const int c = REV ? nchunks - 1 - ci : ci;
            const int t0 = c * K;
            const bool more = ci + 1 < nchunks;
            const int t0_next = more ? t0 + dir * K : t0;   // state rows to prefetch while this chunk runs

            // block set that chunk c+dir needs in addition; its loads are issued one per step below
            // (the last chunk re-reads its own set: harmless, keeps the step body branch-free)
            const int bb_new = more ? (REV ? c - 1 : c + 2) : c;
            // ---- boundary values for the edge lane: broadcast LDS reads, off the dependency chain ----
            u64 bcv[K];
            // fwd: lane 0 at step t0+k needs column t0+k; rev: lane 63 needs column t0+k-63
            const int c_lo = REV ? t0 - 63 : t0;
            auto acquire = [&]() {   // boundary values of the chunk's steps t0 + k, k in [0, K)
                constexpr int K0 = 0, K1 = K;
                int need = 0;  // progress value that guarantees those columns are published
                if (has_pred) {
                    if (REV) {   // (published from the right: the lowest column decides)
                        const int lo = c_lo + K0 < 0 ? 0 : c_lo + K0;
                        need = (c_lo + K1 > 0 && c_lo + K0 < m) ? m - lo : 0;
                    } else {
                        const int hi = c_lo + K1 < m ? c_lo + K1 : m;
                        need = (c_lo + K0 < m) ? hi : 0;
                    }
                }
                if constexpr (PASS == PASS_BWD) {
                    if (need > 0 && imported) {
                        // replay the producer's chunks (it works through the columns from the right, K at a time, and leaves the
                        // progress word at m - first column) down to the one that holds the lowest column needed here
                        const int lo = c_lo < 0 ? 0 : c_lo;
                        slot_t *bnd_w = bnd + (size_t)pslot * p.mcap;
                        while (ximp >= lo / K) {
                            unsigned xg_lo = 0, xg_hi = 0;
                            xb_wait(ximp, c, xg_lo, xg_hi);
                            const int col = ximp * K + lane;
                            if (lane < K && col < m) bnd_w[col] = (slot_t)xg_lo;   // (value in the low word, 0 above it)
                            if (lane == 0) lds_store_i32(prog + 4 * pword, m - ximp * K);
                            --ximp;
                        }
                    }
                }
                if (need > 0) {
                    // bounded spin: a missed hand-off must never hang the device (~0.2 s at the cap).  If it ever
                    // gives up the results of this pair are wrong, so the wave says so in the status words the host
                    // checks on its next call (sdp_api.hip: SDP_E_HANDOFF) -- it does not carry on silently.
                    bool ready = ABL_NOSYNC;
                    const int spin_cap = (SDP_EXP_BUILD && (p.dbg & 8)) ? (1 << 10) : (1 << 21);
                    if constexpr (PIPE && VEC_BND && K0 == 0 && K1 == K && !ABL_NOSYNC) {
                        if (!imported && c_lo >= 0 && c_lo + K <= m) {
                            // ONE LDS round trip: the progress word first, the values right behind it (and both behind the flush's
                            // reads issued at the top of the iteration).  LDS executes a wave's reads in order, so if the word
                            // already covers the chunk -- the steady state: the strip below runs a lag ahead -- the values read
                            // behind it are the published ones; otherwise spin as before and read them again.
                            int seen = lds_issue_i32(prog + 4 * pword);
                            const uint4 *src = reinterpret_cast<const uint4 *>(bnd_in + (c_lo - 1));
                            unsigned w[K + 1];
#pragma unroll
                            for (int g = 0; g < K / 4; ++g) {
                                const uint4 v = src[g];
                                w[4 * g] = v.x, w[4 * g + 1] = v.y, w[4 * g + 2] = v.z, w[4 * g + 3] = v.w;
                            }
                            w[K] = bnd_in[c_lo + K - 1];
                            lds_wait(seen);
                            if (__builtin_amdgcn_readfirstlane(seen) >= pbase + need) {
#pragma unroll
                                for (int k = 0; k < K; ++k) bcv[k] = w[k + 1];
                                return;
                            }
                        }
                    }
                    for (int spin = 0; !ABL_NOSYNC && spin < spin_cap; ++spin) {
                        if (__builtin_amdgcn_readfirstlane(lds_load_i32(prog + 4 * pword)) >= pbase + need) {
                            ready = true;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (!ready && p.status && lane == 0) {
                        if (__hip_atomic_fetch_add(p.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0) {
                            p.status[1] = b, p.status[2] = s, p.status[3] = c | (PASS << 24);
                        }
                    }
                    if constexpr (VEC_BND && K0 == 0 && K1 == K) {
                        if (c_lo >= 0 && c_lo + K <= m) {
                            // columns t0 - 63 .. t0 - 32: the K four-byte slots from the aligned t0 - 64 on as K / 4 16-byte reads, plus one
                            const uint4 *src = reinterpret_cast<const uint4 *>(bnd_in + (c_lo - 1));
                            unsigned w[K + 1];
#pragma unroll
                            for (int g = 0; g < K / 4; ++g) {
                                const uint4 v = src[g];
                                w[4 * g] = v.x, w[4 * g + 1] = v.y, w[4 * g + 2] = v.z, w[4 * g + 3] = v.w;
                            }
                            w[K] = bnd_in[c_lo + K - 1];
#pragma unroll
                            for (int k = 0; k < K; ++k) bcv[k] = w[k + 1];
                            return;
                        }
                    }
                    if (c_lo + K0 >= 0 && c_lo + K1 <= m) {
#pragma unroll
                        for (int k = K0; k < K1; ++k) bcv[k] = bnd_in[c_lo + k];
                    } else {
#pragma unroll
                        for (int k = K0; k < K1; ++k) {
                            const int col = c_lo + k;
                            bcv[k] = (col >= 0 && col < m) ? bnd_in[col] : edge_zero<KIND>();
                        }
                    }
                } else {
#pragma unroll
                    for (int k = K0; k < K1; ++k) bcv[k] = edge_zero<KIND>();
                }
            };
            u64 hist[K];  // the edge-facing carry after each step (published below by one lane)
            const int par = c & 1;
            const int fpar = (t0 / KF) & 1;   // the half of the output ring this chunk's flush block lives in (KF = K: par)
            float *lo = lds_out + lane * PO + (t0 & (2 * KF - 1));  // this lane's row, this chunk's part of the ring
            float *lo_m1 = lds_out + lane * PO - 1;                  // position -1 of the row (FLUSH2)
            const bool top_of_block = ((t0 + K) & (KF - 1)) == 0;    // the chunk holds the last step of a flush block (KF = K: always)
            // ---- publish boundary values for the next strip (one lane), then the progress word ----
            auto publish_range = [&]() {   // values of the chunk's steps t0 + k, k in [0, K)
                constexpr int K0 = 0, K1 = K;
                if (!has_succ) return;
                // fwd: lane 63 produced column t0+k-63 at step k; rev: lane 0 produced column t0+k
                const int p_lo = REV ? t0 : t0 - 63;
                // (reverse sweeps: a chunk that lies right of the matrix for the publishing lane -- the first one or two of every
                //  strip -- has no value to hand on, only its progress word; the column-by-column path below took ~3000 cycles
                //  to find that out, on the strip above's way to its first chunk)
                if (lane == PUB_LANE && !(REV && p_lo + K0 >= m)) {
                    if constexpr (VEC_BND && K0 == 0 && K1 == K) {
                        if (p_lo + K <= m) {   // (p_lo = t0: a multiple of K slots of four bytes)
                            uint4 *dst = reinterpret_cast<uint4 *>(bnd_out + p_lo);
#pragma unroll
                            for (int g = 0; g < K / 4; ++g)
                                dst[g] = make_uint4((unsigned)hist[4 * g], (unsigned)hist[4 * g + 1], (unsigned)hist[4 * g + 2], (unsigned)hist[4 * g + 3]);
                        } else {
#pragma unroll
                            for (int k = 0; k < K; ++k) {
                                const int col = p_lo + k;
                                if (col < m) bnd_out[col] = (slot_t)hist[k];
                            }
                        }
                    } else if (p_lo + K0 >= 0 && p_lo + K1 <= m) {
#pragma unroll
                        for (int k = K0; k < K1; ++k) bnd_out[p_lo + k] = (slot_t)hist[k];
                    } else {
#pragma unroll
                        for (int k = K0; k < K1; ++k) {
                            const int col = p_lo + k;
                            if (col >= 0 && col < m) bnd_out[col] = (slot_t)hist[k];
                        }
                    }
                }
                int done;  // columns published so far (fwd: from the left; rev: from the right)
                if (REV) {
                    done = t0 + K0 < m ? m - (t0 + K0) : 0;
                } else {
                    const int hi = t0 + K1 - 63;
                    done = hi < 0 ? 0 : (hi > m ? m : hi);
                }
                // LDS executes a wave's DS instructions in order, so the data written above is visible to
                // any wave that observes this word (the asm statements also stop compiler reordering)
                if constexpr (PASS == PASS_FWD) {   // (only the ablation build without the recurrence publishes a forward chunk from here)
                    if (lane == PUB_LANE) {
#pragma unroll
                        for (int sb = 0; sb < K / WB; ++sb) frm_out[c * (K / WB) + sb] = FRAME_NONE;
                    }
                }
                if (lane == PUB_LANE && !(SDP_EXP_BUILD && (p.dbg & 8))) lds_store_i32(prog + 4 * oword, obase + done);
                if constexpr (PASS == PASS_BWD) {
                    if (exported && t0 < m) {
                        // the chunk's columns to the bridge row, one granule per lane (tag 0; columns past the matrix: dummies)
                        const int col = t0 + lane;
                        const unsigned gv = (lane < K && col < m) ? (unsigned)bnd_out[col] : 0u;
                        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                        if (!(SDP_EXP_BUILD && (p.dbg & 8)))
                            __builtin_amdgcn_raw_buffer_store_b64((u32x2){gv, 0u}, rs_xo, lane < K ? (unsigned)(col * 8) : OOB, 0, XB_AUX);
                    }
                }
            };
            stamp_rev(0);
            bool zero_chunk = false;   // (ZSKIP) this chunk can only produce +0: see "exact zeros" below
            // +0 bit for bit in every lane's three carries, in the K boundary values and -- where the chunk holds the terminal
            // cell -- in the cotangent
            auto zero_test = [&]() {
                unsigned any = __float_as_uint(cy.fa) | __float_as_uint(cy.fb) | __float_as_uint(cy.fc);
#pragma unroll
                for (int k = 0; k < K; ++k) any |= lo32(bcv[k]);
                if (t_final >= t0 && t_final < t0 + K) any |= __float_as_uint(et);
                return __builtin_amdgcn_ballot_w64(any != 0) == 0 && !(p.flags & 1) && !(SDP_EXP_BUILD && (p.dbg & 4096));
            };
            auto zero_fill_ring = [&]() {   // a zero chunk's outputs: +0 to its half of the ring, its boundary values, the row's copy of step K - 1
#pragma unroll
                for (int k = 0; k < K; ++k) hist[k] = 0;
                if (!((zring >> par) & 1)) {   // (from the third chunk of a run on both halves are zero already)
#pragma unroll
                    for (int k = 0; k < K; ++k) lo[k] = 0.f;
                    zring |= 1 << par;
                }
                if constexpr (FLUSH2) { if (top_of_block) *lo_m1 = 0.f; }   // (shared by both halves)
            };
            if constexpr (PIPE) {
                // the iteration's one wait for memory: this chunk's rows (requested an iteration ago) and whatever is older.  A
                // pipelined chunk issued its 16 output stores BEHIND that request (vmcnt retires in order); every other kind of
                // iteration requested the rows last
                if (tail_st == K / 2) __builtin_amdgcn_s_waitcnt(0x4F70);        // vmcnt(16)
                else if (tail_st == K / 4) __builtin_amdgcn_s_waitcnt(0x0F78);   // vmcnt(8): the eight wide stores of an all-zero flush
                else __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0)
            } else if constexpr (LAZY) {
                // (the fp32 backward sweep waits inside its flush and requests rows further down: see there)
            } else {
            // Everything this wave has in flight is a whole iteration old: the rows loaded at the top of the previous iteration (the
            // ones this chunk consumes) and that iteration's output stores.  Waiting for all of it HERE, with the builtin the
            // compiler's bookkeeping understands, is free -- and leaves the compiler nothing to wait for at the rows' first use.
            // Left to itself it waits there with vmcnt(10): "the ten loads issued since may stay out" -- but it does not count
            // the 16 output stores issued behind those loads, the counter retires in order, and so the ten youngest operations
            // are stores and the wait takes the rows just requested for the NEXT chunk along: the prefetch distance shrinks to
            // the few hundred cycles of the flush.
            if constexpr (ROT) __builtin_amdgcn_s_waitcnt(0x0F70);
            if constexpr (TOPLOAD_X && !ABL_NOLOAD && !ZSKIP_A) {   // (ZSKIP_A: requested further down, once the kind of the next chunk is known)
                if (ci < nchunks) {   // (uniform; the extra flush iterations load nothing)
                    const int c_ = REV ? nchunks - 1 - ci : ci;
                    const int tn = (ci + 1 < nchunks) ? (c_ + dir) * K : c_ * K;   // the last chunk re-reads its own rows: harmless
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        if constexpr (T::QIN == Q_EXACT) rqx2[nxt_t::value][k] = load_f2(rs_qx, tn, k);
                        if constexpr (T::DIN) rdd2[nxt_t::value][k] = load_d(tn, k);
                    }
                }
            }
            if constexpr (TOPLOAD) {
                if (ci < nchunks) {   // (uniform; the extra flush iterations load nothing)
                    const int c_ = nchunks - 1 - ci;
                    const int tn = (ci + 1 < nchunks) ? (c_ - 1) * K : c_ * K;   // the last chunk re-reads its own records: harmless
#pragma unroll
                    for (int jj = 0; jj < QROWS; ++jj) load_q20(tn, jj, nxt_t{});
                }
            }
            }
            float2 fv[PIPE ? K / 2 : 1];   // (PIPE) the previous chunk's outputs, out of LDS and on their way to memory
            bool fl_defer = false;         // (PIPE) ... their stores have not been issued yet
            const bool fl_zero = ZSKIP && zring == 3;   // ... and they are all +0 (the last two chunks were zero chunks)
            if constexpr (T::SOUT > 0) {
                if constexpr (PIPE) {
                    fl_defer = ci > 0 && rows == 64 && pf_t0 >= K && pf_t0 + K <= m;   // the plain flush: every element a real cell
                    if (fl_defer) flush_read(pf_par, fl_zero, fv);
                    else {
                        flush_out(pf_t0, pf_par, ci > 0, ZSKIP && zring == 3);
#pragma unroll
                        for (int k2 = 0; k2 < K / 2; ++k2) fv[k2] = make_float2(0.f, 0.f);
                    }
                } else {
                    // (KF > K: a flush block is complete after every second chunk)
                    if (KF == K || (pf_t0 & (KF - 1)) == 0) flush_out(pf_t0, pf_par, ci > 0, ZSKIP && zring == 3);
                }
                stamp_rev(1);
                if (ci >= nchunks) {
                    if constexpr (PIPE) {
                        if (fl_defer) {
                            if (fl_zero) flush_zero8(pf_t0);
                            else {
#pragma unroll
                                for (int k2 = 0; k2 < K / 2; ++k2) flush_store1(pf_t0, k2, fv[k2], (unsigned)(f2_g * 4));
                            }
                        }
                        tail_st = 0;
                    }
                    pf_t0 = -K, pf_par = 1;
                    return;
                }
            }
            bool za_acquired = false;
            if constexpr (ZSKIP_A) {
                using ecur_t = std::integral_constant<int, P>;       // the E block set that arrived for the NEXT chunk (c - 1) ...
                using enxt_t = std::integral_constant<int, 1 - P>;   // ... and the set the one after it is loaded into
                const bool zb_new = block_zero(ecur_t{});
                // "Zero" for the float64 carries means NEGLIGIBLE: below 2^-170 in magnitude.  The carries of this sweep are float64 and
                // do not underflow where the fp32 E did; but with e = 0 over the chunk they only get redistributed by weights that
                // sum to 1, so if everything that enters the chunk is below 2^-170 every Ed in it is below 2^-162 -- and is stored as
                // the float 0 (the smallest fp32 denormal is 2^-149).  THE RULE, part of the algorithm in both modes: a chunk with
                // E = 0 whose incoming carries and boundary values are all below 2^-170 starts from exact zeros (what that drops from
                // any later Ed is below 2^-162, a 2^-13th of the last bit of the smallest denormal).  With the rule applied either
                // way, skipping such a chunk or computing it (SDP_NO_ZERO_SKIP) gives the same bits.
                constexpr unsigned TINY = (1023u - 170u) << 20;   // high word of 2^-170
                auto mag64 = [](u64 v) -> unsigned { return hi32(v) & 0x7fffffffu; };   // < TINY iff |v| < 2^-170
                bool zc = known_zero;
                if (!zc) {
                    acquire();
                    za_acquired = true;
                    unsigned big = max(max(mag64((u64)__double_as_longlong(cy.a)), mag64((u64)__double_as_longlong(cy.b))), mag64((u64)__double_as_longlong(cy.c)));
#pragma unroll
                    for (int k = 0; k < K; ++k) big = max(big, mag64(bcv[k]));
                    zc = za_b1 && za_b2 && __builtin_amdgcn_ballot_w64(big >= TINY) == 0;
                    if (zc) {   // the rule: from exact zeros
                        cy.a = cy.b = cy.c = 0.0;
#pragma unroll
                        for (int k = 0; k < K; ++k) bcv[k] = 0;
                    }
                }
                const bool zskip = zc && !(p.flags & 1);   // (SDP_NO_ZERO_SKIP: the chunk is computed all the same -- from the same zeros)
                // will the next chunk be one too?  Its carries stay zero if this one is; its E is known (zb_new, za_b1); its
                // boundary values only if the strip below has published them already (not waited for)
                bool nz = zskip && more && zb_new && za_b1;
                if (nz) {
                    unsigned any = 0;
                    if (has_pred) {
                        const int c_lo_n = t0_next - 63;
                        const int need = (c_lo_n + K > 0 && c_lo_n < m) ? m - (c_lo_n < 0 ? 0 : c_lo_n) : 0;
                        if (imported || (need > 0 && __builtin_amdgcn_readfirstlane(lds_load_i32(prog + 4 * pword)) < pbase + need)) any = 1;
                        else {
#pragma unroll
                            for (int k = 0; k < K; ++k) {
                                const int col = c_lo_n + k;
                                if (col >= 0 && col < m) any |= mag64((u64)bnd_in[col]) >= TINY ? 1u : 0u;
                            }
                        }
                    }
                    nz = __builtin_amdgcn_ballot_w64(any != 0) == 0;
                }
                // the next chunk's Q and Qd rows -- or nothing (a descriptor of zero records returns zeros without touching memory)
                {
                    const int tn = more ? t0_next : t0;
                    const bool wanted = more && !nz;
                    const __amdgpu_buffer_rsrc_t rq_ = make_rsrc(reinterpret_cast<const char *>(p.qin) + ps_idx * p.st2_ps, wanted ? ST_RECORDS : 0u);
                    const __amdgpu_buffer_rsrc_t rd_ = make_rsrc(reinterpret_cast<const char *>(p.din) + ps_idx * p.st2_ps, wanted ? ST_RECORDS : 0u);
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        rqx2[nxt_t::value][k] = load_f2(rq_, tn, k);
                        rdd2[nxt_t::value][k] = load_f2(rd_, tn, k);
                    }
                }
                load_block_s((ci + 2 < nchunks) ? c - 2 : c - 1, enxt_t{});   // E two chunks ahead
                za_b2 = za_b1, za_b1 = zb_new;
                known_zero = nz;
                if (zskip) {
#pragma unroll
                    for (int k = 0; k < K; ++k) hist[k] = 0, lo[k] = 0.f;
                    if constexpr (FLUSH2) { if (top_of_block) *lo_m1 = 0.f; }
                    cy.a = cy.b = cy.c = 0.0;
                    stamp_rev(2), stamp_rev(3);
                    publish_range();
                    stamp_rev(4);
                    pf_t0 = t0, pf_par = fpar;
                    if (more) write_block_s(bb_new, ecur_t{});
                    return;
                }
            }
            const bool pipe_interior = PIPE && chunk_interior(c);   // this chunk's steps will run in the pipelined body
            bool pipe_now = false;                                    // ... and carry the previous chunk's stores
            if constexpr (LAZY) {
                // The flush above ended with a wait for everything older than its own stores (flush_out, LAZY): this chunk's rows,
                // requested an iteration ago, and that iteration's outputs -- free, and said with the builtin the compiler's
                // bookkeeping understands, so that it has nothing to add at the rows' first use (a wait it places there takes the
                // rows just requested for the NEXT chunk along).  Then: what kind of chunk is this one (known from the iteration
                // before, or found out now from its boundary values), and -- if it is a zero chunk, whose carries stay +0 --
                // will the next one be too?  Its rows are requested accordingly, at this one place, and a zero chunk leaves the
                // iteration: no path from it reaches the steps.
                static_assert(!LAZY || (T::SIN == 0 && T::SOUT > 0), "the zero-chunk path below stages no inputs");
                if (known_zero) {
                    zero_chunk = true;
                } else {
                    acquire();
                    zero_chunk = zero_test();
                }
                stamp_rev(2);
                const bool next_zero = zero_chunk && more && boundary_zero(t0_next, false);
                if constexpr (PIPE) pipe_now = fl_defer && !zero_chunk && pipe_interior;
                load_rows(t0_next, nxt_t{}, more && !next_zero);
                known_zero = next_zero;
                if constexpr (PIPE) {
                    tail_st = (!zero_chunk && pipe_interior) ? K / 2 : 0;   // (a pipelined body issues its 16 stores whether or not they are dummies)
                    if (fl_defer && !pipe_now) {
                        // no pipelined steps to carry them: the stores leave here, BEHIND the request for rows -- the next iteration's
                        // wait then leaves them outstanding (in a run of zero chunks nothing ever waits for a store's acknowledgement);
                        // an all-zero flush is eight wide stores out of no registers at all
                        if (fl_zero) flush_zero8(pf_t0), tail_st += K / 4;
                        else {
#pragma unroll
                            for (int k2 = 0; k2 < K / 2; ++k2) flush_store1(pf_t0, k2, fv[k2], (unsigned)(f2_g * 4));
                            tail_st += K / 2;
                        }
                    }
                }
                if (zero_chunk) {
                    zero_fill_ring();
                    stamp_rev(3);
                    publish_range();
                    stamp_rev(4);
                    pf_t0 = t0, pf_par = fpar;
                    return;
                }
            }
            // experiments build, sdp_set_trace: stamps 4..7 of a chunk's two block slots bracket the chunk's memory pipeline
            auto stamp_chunk = [&](int blk, int k) {
                if constexpr (SDP_EXP_BUILD != 0 && PASS == PASS_FWD) {
                    if (p.trace && !(p.dbg & 1024) && (b & 63) == 0 && b < 256 && lane == 0 && (parts ? part < 4 : sidx / W < 2) && blk < 40)
                        p.trace[((((b >> 6) * 4 + wave) * 4 + (parts ? part : sidx / W)) * 40 + blk) * 8 + k] = __builtin_readcyclecounter();
                }
            };
            stamp_chunk(t0 / WB, 4);
            if constexpr (!ZSKIP_A) load_block(bb_new);   // (ZSKIP_A: requested a chunk earlier, see there)
            stamp_chunk(t0 / WB, 5);

            if constexpr (FWD_SUB) {
                fwd_blocks(c, t0);
                stamp_chunk(t0 / WB + 1, 4);
                if (more) write_block(bb_new);
                stamp_chunk(t0 / WB + 1, 5);
                return;
            }

            if constexpr (ZSKIP_A) { if (!za_acquired) acquire(); }
            else if constexpr (!LAZY) acquire();   // (LAZY: further up, in front of the decision what kind of chunk this is)
            if constexpr (!LAZY) stamp_rev(2);

            // ---- staged inputs of this chunk: one burst of LDS reads, off the dependency chain ----
            float in0[K], in1[K], in2[T::SIN > 2 ? K : 1];
            if constexpr (T::SIN > 0) {
                if constexpr (ABL_NOLDS) {
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        in0[k] = rs[0][0][k];
                        if constexpr (T::SIN > 1) in1[k] = rs[0][1][k];
                        if constexpr (T::SIN > 2) in2[k] = rs[0][2][k];
                    }
                } else {
                    const int pr = (t0 & (RING - 1)) + 4 * ring_pi(lane & 7);  // ring position of step t0 for this lane
#pragma unroll
                    for (int g = 0; g < K / 4; ++g) {
                        const int idx = lane * PITCH + ((pr + 4 * g) & (RING - 1));
                        const float4 v0 = *reinterpret_cast<const float4 *>(lds_in + idx);
                        in0[4 * g] = v0.x, in0[4 * g + 1] = v0.y, in0[4 * g + 2] = v0.z, in0[4 * g + 3] = v0.w;
                        if constexpr (T::SIN > 1) {
                            const float4 v1 = *reinterpret_cast<const float4 *>(lds_in + PLANE + idx);
                            in1[4 * g] = v1.x, in1[4 * g + 1] = v1.y, in1[4 * g + 2] = v1.z, in1[4 * g + 3] = v1.w;
                        }
                        if constexpr (T::SIN > 2) {
                            const float4 v2 = *reinterpret_cast<const float4 *>(lds_in + 2 * PLANE + idx);
                            in2[4 * g] = v2.x, in2[4 * g + 1] = v2.y, in2[4 * g + 2] = v2.z, in2[4 * g + 3] = v2.w;
                        }
                    }
                }
            }

            const bool interior = chunk_interior(c);
            // The fp32 backward sweep needs no masks in the RAMPS of a full strip either (round 5; the masked body costs ~20
            // instead of ~14 instructions per step, and a strip's first three chunks -- all ramp -- are what the strip above
            // waits for before it can start: three such lags are a quarter of the launch at 256 x 512^2):
            //   * right of the matrix (col >= m, the start of the sweep) a cell receives only from cells further right or below,
            //     all outside as well, and from boundary values that `acquire` hands over as +0 for columns outside: with the
            //     strip's carries starting at +0 it computes e = 0 + 0 and hands on q * 0 = +0 (the packed weights of cells
            //     outside the matrix are finite numbers) -- exactly what the mask would have put there;
            //   * left of the matrix (col < 0, the end of the sweep) a cell computes garbage, but hands it only to cells that are
            //     further left, never to one inside; its output is not stored (every flush that can meet such a column tests
            //     it), its boundary values are not published (columns >= 0 only), and the strip's carries die with the strip.
            // Not for: the chunk that holds the terminal cell (its cotangent is injected under the mask), partial strips (rows
            // below the matrix do hand upwards), Smith-Waterman's border row and column, float2 states (records of cells outside
            // may hold NaN).
            const bool mask_free = interior || (PASS == PASS_BWD && T::QIN == Q_PACKED && rows == 64 && !sw &&
                                                !(s == nstrips - 1 && m - 1 + 63 >= t0 && m - 1 + 63 < t0 + K));

            // ---- K steps (every pass but the forward sweep, which runs fwd_blocks); EDGE=false is the mask-free body for chunks fully inside the matrix ----
            const unsigned pipe_off = pipe_now ? (unsigned)(f2_g * 4) : OOB;   // (PIPE) per-lane offset of the stores the steps carry (OOB: dummies)
            auto steps = [&](auto edge_tag, auto pipe_tag) {
                constexpr bool EDGE = decltype(edge_tag)::value;
                constexpr bool PIPED = decltype(pipe_tag)::value;   // the previous chunk's stores ride along
#pragma unroll
                for (int kk = 0; kk < K; ++kk) {
                    const int k = REV ? K - 1 - kk : kk;
                    const int t = t0 + k;
                    const int col = t - lane;
                    const bool inside = !EDGE || (unsigned)col < (unsigned)m;
                    const bool dead = EDGE && sw && (col == 0 || (i0 + lane) == 0);  // SW: padded row 1 / col 1
                    const bool rowok = !EDGE || lane < rows;

                    float2 q0, q1;
                    if constexpr (T::QIN == Q_EXACT) q0 = rqx2[P][k];
                    if constexpr (T::QIN == Q_PACKED) {
                        unsigned w5[5];
                        q20_record(k >> 2, w5, cur_t{});
                        q0 = q20_unpack(w5, k & 3);
                        q0.x *= QF_UNSCALE, q0.y *= QF_UNSCALE;
                    }
                    if constexpr (T::DIN) q1 = rdd2[P][k];

                    if constexpr (ABL_NOMATH) {
                        if constexpr (T::QOUT != Q_NONE || T::DOUT) {
                            float2 qq = make_float2(in0[k], T::QIN != Q_NONE ? q0.x + q0.y : in1[k]);
                            if constexpr (ABL_NOSTORE) { keep(qq.x); keep(qq.y); } else store_state(t0, k, qq);
                        }
                        if constexpr (T::SOUT > 0) lo[k] = q0.x + q0.y + (T::DIN ? q1.x + q1.y : 0.f) + (T::SIN > 0 ? in0[k] : 0.f);
                        hist[k] = 0;
                    } else if constexpr (PASS == PASS_AFWD) {
                        // Fused loss seed: Ztheta = dLoss/dE is formed here from the loss's own operands --
                        // ref (Ytrue or the path matrix), pred (= E, the alignment matrix the loss was evaluated on), the
                        // mask G and the per-pair factor -- instead of being written by the loss's backward kernel and read
                        // back (deepblast/losses.py:26-46, 69-79, 108-118; same formulas as sdp_loss_bwd_kernel).  Cells
                        // outside the pair's block are outside the matrix here (`inside`), so no extra length test.
                        float zt, za;
                        if constexpr (QX) {
                            zt = (in2[k] != 0.f && inside) ? loss_dterm(in0[k], in1[k], seed_scale, p.loss_kind) : 0.f;
                            za = 0.f;
                        } else {
                            zt = in0[k];
                            za = in1[k];
                        }
                        const double up = dpp_f64<DPP_IN>(__longlong_as_double((long long)bcv[k]), cy.a);
                        const double diag = cy.b, left = cy.a;
                        const bool live = inside && !dead;
                        const double qx = live ? (double)q0.x : 0.0, qy = live ? (double)q0.y : 0.0;
                        const double qm = live ? (1.0 - qx) - qy : 0.0;
                        const double zad = (double)za;
                        const double a0 = zad + up, a1 = diag, a2 = zad + left;
                        const double tot = __builtin_fma(qy, a2, __builtin_fma(qm, a1, qx * a0));
                        const double vd = (double)zt + tot;
                        {
                            float2 qq = make_float2((float)(qx * (a0 - tot)), (float)(qy * (a2 - tot)));
                            if constexpr (ABL_NOSTORE) { keep(qq.x); keep(qq.y); } else store_state(t0, k, qq);
                        }
                        cy.b = up;
                        cy.a = inside ? vd : 0.0;
                        hist[k] = (u64)__double_as_longlong(cy.a);
                        if constexpr (EDGE) vt_keep = (t == t_final) ? hist[k] : vt_keep;
                    } else if constexpr (PASS == PASS_BWD) {
                        const float in = __uint_as_float(dpp_i32<DPP_IN>((int)lo32(bcv[k]), __float_as_int(cy.fa)));
                        float e = in + cy.fb;
                        if constexpr (EDGE) {
                            // one unsigned compare per step: the lane's cell is real and not a Smith-Waterman
                            // border cell iff live_lo <= t < live_lo + live_span (see the strip setup)
                            const bool live = (unsigned)(t - live_lo) < live_span;
                            e = (t == t_final) ? et : e;
                            e = live ? e : 0.f;
                        }
                        // the packed weights need no masking: the fields decode to finite numbers, and a cell that
                        // is not live hands on e = 0; float2 records of dead cells may hold anything (NaN included)
                        float qx = q0.x, qy = q0.y;
                        if constexpr (EDGE && T::QIN == Q_EXACT) {
                            const bool live = (unsigned)(t - live_lo) < live_span;
                            qx = live ? qx : 0.f, qy = live ? qy : 0.f;
                        }
                        const float qm = __builtin_fmaxf((1.f - qx) - qy, 0.f);  // the two stored weights are rounded independently
                        cy.fb = qy * e;
                        cy.fa = __builtin_fmaf(qx, e, cy.fc);  // px + pm of the previous step
                        cy.fc = qm * e;
                        lo[k] = e;
                        hist[k] = (u64)__float_as_uint(cy.fa);
                        if constexpr (PIPED) {
                            // one of the previous chunk's 16 output stores per step of the chunk's first half
                            if (kk < K / 2) flush_store1(pf_t0, kk, fv[kk], pipe_off);
                        }
                    } else {  // PASS_ABWD
                        const float ef = in0[k];
                        const double in = dpp_f64<DPP_IN>(__longlong_as_double((long long)bcv[k]), cy.a);
                        const bool cell = inside && rowok;
                        const bool live = cell && !dead;
                        const double ed = cell ? in + cy.b : 0.0;
                        const double e = live ? (double)ef : 0.0;
                        const double qx = live ? (double)q0.x : 0.0, qy = live ? (double)q0.y : 0.0;
                        const double qm = live ? (1.0 - qx) - qy : 0.0;
                        const double dx = live ? (double)q1.x : 0.0, dy = live ? (double)q1.y : 0.0;
                        const double dm = -(dx + dy);
                        cy.b = __builtin_fma(dy, e, qy * ed);
                        cy.a = __builtin_fma(dx, e, __builtin_fma(qx, ed, cy.c));  // gx + gm of the previous step
                        cy.c = __builtin_fma(dm, e, qm * ed);
                        lo[k] = ZSKIP_A ? (float)ed + 0.0f : (float)ed;   // (ZSKIP_A: no -0 in Ed, whichever way a zero came about)
                        hist[k] = (u64)__double_as_longlong(cy.a);
                    }
                }
            };
            // ---- exact zeros (fp32 backward sweep) ----
            // E is a sum of products of weights along paths: away from the alignment it underflows to exactly +0 (47 % of the
            // cells of the benchmark's soft scores, nearly all of a peaked alignment).  If every lane's three carries, the K
            // boundary values from the strip below and (where the chunk holds the terminal cell) the cotangent are +0 bit for
            // bit, every step of the chunk computes e = 0 + 0, q * 0 and fma(q, 0, 0) with finite q >= 0: +0 outputs, +0
            // carries, +0 values for the strip above -- so those are written without running the steps.  (-0, which a
            // negative cotangent leaves behind, is not taken: sums of mixed zeros depend on the order.)
            if constexpr (ZSKIP) {
                if constexpr (!LAZY) zero_chunk = zero_test();   // (LAZY: a zero chunk never gets here)
                if (zero_chunk) zero_fill_ring();
                else zring &= ~(1 << par);
            }
            if (!zero_chunk) {
                if constexpr (PIPE) {
                    if (interior) steps(std::false_type{}, std::true_type{});
                    else if (mask_free) steps(std::false_type{}, std::false_type{});
                    else steps(std::true_type{}, std::false_type{});
                } else {
                    if (mask_free) steps(std::false_type{}, std::false_type{});
                    else steps(std::true_type{}, std::false_type{});
                }
            }
            stamp_rev(3);
            publish_range();
            if constexpr (FLUSH2) {
                if (!zero_chunk && top_of_block) *lo_m1 = lo[K - 1];   // step 31's value once more, at position -1 of the row (see FLUSH2)
            }
            stamp_rev(4);

            // ---- flush: one memory-aligned K-element block per row (see fo_* above) ----
            pf_t0 = t0, pf_par = fpar;

            if constexpr (ZSKIP_A) { if (more) write_block_s(bb_new, std::integral_constant<int, P>{}); }
            else { if (more) write_block(bb_new); }
        };
        if constexpr (ROT) {
            for (int ci = 0; ci < nchunks + nflush; ci += 2) {
                chunk_body(std::integral_constant<int, 0>{}, ci);
                if (ci + 1 < nchunks + nflush) chunk_body(std::integral_constant<int, 1>{}, ci + 1);
            }
        } else {
            for (int ci = 0; ci < nchunks + nflush; ++ci) chunk_body(std::integral_constant<int, 0>{}, ci);
        }
        if constexpr (!REV) {
            if (t_final >= 0) {
                if constexpr (KIND == CK_EXP) {
                    // V = e*ln2 + ln(a), once per pair, in float64
                    p.vout[b] = (float)((double)(int)hi32(vt_keep) * 0.69314718055994530942 +
                                        log((double)__uint_as_float(lo32(vt_keep))));
                } else {
                    p.vout[b] = (float)__longlong_as_double((long long)vt_keep);
                }
            }
        }
    }
    zero_fill();  // every wave is busy (no idle ones): each clears its share once its strips are done
}

}  // namespace sdp

// ----------------------------------------------------------------------------------
// kernels (one symbol per pass so that rocprofv3 names them)
// ----------------------------------------------------------------------------------
#define SDP_KERNEL(NAME, PASS, K, MAXW, ...)                                               \
    extern "C" __global__ void __launch_bounds__((MAXW) * 64) NAME(const sdp::Params p)    \
    {                                                                                      \
        sdp::sweep<PASS, K, ##__VA_ARGS__>(p);                                             \
    }

// (-DSDP_ONLY=<n>: compile ONE kernel, for ISA inspection with `hipcc -S --cuda-device-only` -- tools/isa.sh; never linked)
#if defined(SDP_ONLY) && SDP_ONLY == 1
SDP_KERNEL(sdp_bwd_kernel, sdp::PASS_BWD, SDP_K_BWD, SDP_MAXW_BWD_Q, false, false, false, false, true)
#elif defined(SDP_ONLY) && SDP_ONLY == 21
SDP_KERNEL(sdp_bwd_pipe_kernel, sdp::PASS_BWD, SDP_K_BWD, SDP_MAXW_BWD_Q)
#elif defined(SDP_ONLY) && SDP_ONLY == 11
SDP_KERNEL(sdp_bwd_g_kernel, sdp::PASS_BWD, SDP_K_BWD, SDP_MAXW_BWD_Q, false, false, true)
#elif defined(SDP_ONLY) && SDP_ONLY == 18
SDP_KERNEL(sdp_bwd_x_g_kernel, sdp::PASS_BWD, SDP_K_BWD, SDP_MAXW_BWD, true, false, true)
#elif defined(SDP_ONLY) && SDP_ONLY == 15
SDP_KERNEL(sdp_bwd_lat_g_kernel, sdp::PASS_BWD, SDP_K_BWD_LAT, SDP_MAXW_BWD_LAT, false, false, true)
#elif defined(SDP_ONLY) && SDP_ONLY == 0
SDP_KERNEL(sdp_fwd_kernel, sdp::PASS_FWD, SDP_K_FWD, SDP_MAXW_FWD, false, true, false, false, false, true)
#elif defined(SDP_ONLY) && SDP_ONLY == 37
SDP_KERNEL(sdp_fwd_c_kernel, sdp::PASS_FWD, SDP_K_FWD, SDP_MAXW_FWD, false, true)
#elif defined(SDP_ONLY) && SDP_ONLY == 6
SDP_KERNEL(sdp_fwd_lat_kernel, sdp::PASS_FWD, SDP_K_FWD_LAT, SDP_MAXW_FWD_LAT, false, false, false, false, false, true)
#elif defined(SDP_ONLY) && SDP_ONLY == 9
SDP_KERNEL(sdp_fwd_x_tp_kernel, sdp::PASS_FWD, SDP_K_FWD, SDP_MAXW_FWD, true, true, false, false, false, true)
#elif defined(SDP_ONLY) && SDP_ONLY == 7
SDP_KERNEL(sdp_bwd_x_kernel, sdp::PASS_BWD, SDP_K_BWD, SDP_MAXW_BWD, true)
#elif defined(SDP_ONLY) && SDP_ONLY == 3
SDP_KERNEL(sdp_adj_bwd_kernel, sdp::PASS_ABWD, SDP_K_ABWD, SDP_MAXW_ABWD)
#elif defined(SDP_ONLY) && SDP_ONLY == 2
SDP_KERNEL(sdp_adj_fwd_kernel, sdp::PASS_AFWD, SDP_K_AFWD, SDP_MAXW_AFWD)
#else
// (-DSDP_GROUP=<g>: compile one group of kernels -- deepblast_amd/build.py builds the groups of this file in parallel and links
//  them; without it, everything.  Group 0 holds the small kernels at the end of the file.)
#ifndef SDP_GROUP
#define SDP_GROUP (-1)
#endif
#define SDP_IN_GROUP(g) (SDP_GROUP < 0 || SDP_GROUP == (g))
// template arguments after MAXW: QX (exact state / fused loss seed), LINES (throughput forward builds: line-aligned input blocks),
// GEN (general pitch), PARTS (a pair over several workgroups), NOPIPE (packed backward sweep without the pipelined chunk)
// (... NOCLEAN: the aligned throughput forward builds without the edge cleaning -- full strips, no per-pair lengths; their _c twins carry it)
#if SDP_IN_GROUP(1)
SDP_KERNEL(sdp_fwd_kernel, sdp::PASS_FWD, SDP_K_FWD, SDP_MAXW_FWD, false, true, false, false, false, true)
SDP_KERNEL(sdp_fwd_lat_kernel, sdp::PASS_FWD, SDP_K_FWD_LAT, SDP_MAXW_FWD_LAT, false, false, false, false, false, true)
SDP_KERNEL(sdp_fwd_x_kernel, sdp::PASS_FWD, SDP_K_FWD_LAT, SDP_MAXW_FWD_LAT, true, false, false, false, false, true)
SDP_KERNEL(sdp_fwd_x_tp_kernel, sdp::PASS_FWD, SDP_K_FWD, SDP_MAXW_FWD, true, true, false, false, false, true)
#endif
#if SDP_IN_GROUP(8)
SDP_KERNEL(sdp_fwd_c_kernel, sdp::PASS_FWD, SDP_K_FWD, SDP_MAXW_FWD, false, true)
SDP_KERNEL(sdp_fwd_x_tp_c_kernel, sdp::PASS_FWD, SDP_K_FWD, SDP_MAXW_FWD, true, true)
SDP_KERNEL(sdp_fwd_lat_c_kernel, sdp::PASS_FWD, SDP_K_FWD_LAT, SDP_MAXW_FWD_LAT)
SDP_KERNEL(sdp_fwd_x_c_kernel, sdp::PASS_FWD, SDP_K_FWD_LAT, SDP_MAXW_FWD_LAT, true)
#endif
#if SDP_IN_GROUP(2)
// The packed backward sweep twice: with the chunk as ONE software pipeline (PIPE: deferred flush, stores inside the steps) for
// long pairs on one wave per SIMD, and without it for everything else -- steady-state A/B, fwd;bwd us, +- 0.3 (tools/steady.py,
// profiles/r05_steady_pipe.txt; no pipeline -> pipeline): 256 x 1024^2 1064.5 -> 1036.2, 256 x 512 x 1024 665.5 -> 650.7, 256 x 768 x 640
// 576.1 -> 569.3, 256 x 1024 x 512 602.6 -> 596.6; but 256 x 512^2 272.5 -> 279.0, 64 x 512^2 218.3 -> 224.2, 128 x 512^2 243.3 -> 249.1,
// 512 x 256^2 163.4 -> 169.6, 128 x 1024^2 (8 waves) 816.3 -> 825.1.  sdp_api.hip plan() picks (bwd_pipe_pays).
SDP_KERNEL(sdp_bwd_kernel, sdp::PASS_BWD, SDP_K_BWD, SDP_MAXW_BWD_Q, false, false, false, false, true)
SDP_KERNEL(sdp_bwd_pipe_kernel, sdp::PASS_BWD, SDP_K_BWD, SDP_MAXW_BWD_Q)
SDP_KERNEL(sdp_bwd_lat_kernel, sdp::PASS_BWD, SDP_K_BWD_LAT, SDP_MAXW_BWD_LAT)
SDP_KERNEL(sdp_bwd_x_kernel, sdp::PASS_BWD, SDP_K_BWD, SDP_MAXW_BWD, true)
SDP_KERNEL(sdp_bwd_x_lat_kernel, sdp::PASS_BWD, SDP_K_BWD_LAT, SDP_MAXW_BWD_LAT, true)
#endif
#if SDP_IN_GROUP(3)
SDP_KERNEL(sdp_adj_bwd_g_kernel, sdp::PASS_ABWD, SDP_K_ABWD, SDP_MAXW_ABWD, false, false, true)
SDP_KERNEL(sdp_adj_fwd_kernel, sdp::PASS_AFWD, SDP_K_AFWD, SDP_MAXW_AFWD)
SDP_KERNEL(sdp_adj_fwd_loss_kernel, sdp::PASS_AFWD, SDP_K_AFWD, 4, true)
SDP_KERNEL(sdp_adj_bwd_kernel, sdp::PASS_ABWD, SDP_K_ABWD, SDP_MAXW_ABWD)
#endif
// PARTS instantiations: the throughput builds with the bridge between workgroups (a pair spread over several CUs)
#if SDP_IN_GROUP(4)
SDP_KERNEL(sdp_fwd_p_kernel, sdp::PASS_FWD, SDP_K_FWD, SDP_MAXW_FWD, false, true, false, true)
SDP_KERNEL(sdp_fwd_x_tp_p_kernel, sdp::PASS_FWD, SDP_K_FWD, SDP_MAXW_FWD, true, true, false, true)
SDP_KERNEL(sdp_fwd_pg_kernel, sdp::PASS_FWD, SDP_K_FWD, SDP_MAXW_FWD, false, true, true, true)
SDP_KERNEL(sdp_fwd_x_tp_pg_kernel, sdp::PASS_FWD, SDP_K_FWD, SDP_MAXW_FWD, true, true, true, true)
#endif
#if SDP_IN_GROUP(5)
SDP_KERNEL(sdp_bwd_p_kernel, sdp::PASS_BWD, SDP_K_BWD, SDP_MAXW_BWD, false, false, false, true)
SDP_KERNEL(sdp_bwd_x_p_kernel, sdp::PASS_BWD, SDP_K_BWD, SDP_MAXW_BWD, true, false, false, true)
SDP_KERNEL(sdp_bwd_pg_kernel, sdp::PASS_BWD, SDP_K_BWD, SDP_MAXW_BWD, false, false, true, true)
SDP_KERNEL(sdp_bwd_x_pg_kernel, sdp::PASS_BWD, SDP_K_BWD, SDP_MAXW_BWD, true, false, true, true)
#endif
// general-pitch instantiations (GEN = true) of the kernels that stage outputs, and of the line-aligned forward builds
#if SDP_IN_GROUP(6)
SDP_KERNEL(sdp_fwd_g_kernel, sdp::PASS_FWD, SDP_K_FWD, SDP_MAXW_FWD, false, true, true)
SDP_KERNEL(sdp_fwd_x_tp_g_kernel, sdp::PASS_FWD, SDP_K_FWD, SDP_MAXW_FWD, true, true, true)
#endif
#if SDP_IN_GROUP(7)
SDP_KERNEL(sdp_bwd_g_kernel, sdp::PASS_BWD, SDP_K_BWD, SDP_MAXW_BWD_Q, false, false, true)
SDP_KERNEL(sdp_bwd_lat_g_kernel, sdp::PASS_BWD, SDP_K_BWD_LAT, SDP_MAXW_BWD_LAT, false, false, true)
SDP_KERNEL(sdp_bwd_x_g_kernel, sdp::PASS_BWD, SDP_K_BWD, SDP_MAXW_BWD, true, false, true)
SDP_KERNEL(sdp_bwd_x_lat_g_kernel, sdp::PASS_BWD, SDP_K_BWD_LAT, SDP_MAXW_BWD_LAT, true, false, true)
#endif
#endif
#ifndef SDP_IN_GROUP
#define SDP_IN_GROUP(g) 1
#endif
#if SDP_IN_GROUP(0)

// ----------------------------------------------------------------------------------
// launch order for variable-length batches: order[r] = the pair with the r-th largest n*m (ties: lower index first).
// Rank by counting -- B is at most a few thousand, the (B,2) lengths sit in L2 -- so no sort, no scratch memory.
// ----------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(256) sdp_order_kernel(const int *lens, int *order, int B, int N, int M)
{
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    auto work = [&](int i) {
        int n = lens[2 * i], m = lens[2 * i + 1];
        n = n < 1 ? 1 : (n > N ? N : n);
        m = m < 1 ? 1 : (m > M ? M : m);
        return n * m;
    };
    const int mine = work(b);
    int rank = 0;
    for (int i = 0; i < B; ++i) {
        const int w = work(i);
        rank += (w > mine || (w == mine && i < b)) ? 1 : 0;
    }
    order[rank] = b;
}

// ----------------------------------------------------------------------------------
// bridge rows of a parts launch: every granule "not written yet" (sdp_kernels.h: XB_INVALID in both words)
// ----------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(256) sdp_bridge_reset_kernel(unsigned long long *xb, size_t n8)
{
    const unsigned long long pattern = ((unsigned long long)sdp::XB_INVALID << 32) | sdp::XB_INVALID;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) xb[i] = pattern;
}

// ----------------------------------------------------------------------------------
// dispatch order when pairs are spread over several workgroups ("parts") and have their own lengths: map[h] = pair *
// nparts_max + part for workgroup h.  Workgroups are handed to CUs in index order as CUs free up, and a part that is on
// a CU before its producer has reached it only waits there.  So: every pair's part 0 first, then the parts 1, ... -- part
// k has nothing to do for the first k * 4 * 79 steps (~60 us each) of its pair, about the time the shortest pairs of the
// batch take to leave their CUs -- and within one k by the critical path that still hangs on the part, longest first
// ((P - 1 - k) * 4 * 79 + 3 * 79 + m + 63 steps for a pair of P parts and m columns).  A producer always precedes its
// consumer, so a waiting part never keeps its producer off the chip.  (Ranking by the critical path alone put all parts
// of the long pairs on CUs at once, most of them waiting: forward sweep of BASELINE configs[2] 600 us instead of 511.)
// Slots k of pairs with k parts or fewer come at the end of the parts k: in the forward sweep they exit at once, in the
// backward sweep they zero-fill E outside the pair's block (sorted behind everything else the fill ran at the very end,
// on the few CUs that were free: 483 instead of 3xx us).  Rank by counting, as above.
// ----------------------------------------------------------------------------------
extern "C" __global__ void __launch_bounds__(256) sdp_parts_map_kernel(const int *lens, int *map, int B, int N, int M, int nparts_max, int strips)
{
    // one wavefront per workgroup-to-be: its 64 lanes share the scan over the list (ranking 1024 parts with one thread
    // each took ~50 us -- a tenth of the sweep it was meant to speed up)
    const int h = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int total = B * nparts_max;
    if (h >= total) return;
    auto key = [&](int e) {
        const int pr = e / nparts_max, k = e % nparts_max;
        int n = lens[2 * pr], m = lens[2 * pr + 1];
        n = n < 1 ? 1 : (n > N ? N : n);
        m = m < 1 ? 1 : (m > M ? M : m);
        const int np = ((n + 63) / 64 + strips - 1) / strips;
        return (nparts_max - k) * 65536 + (k < np ? (np - 1 - k) * strips * 79 + (strips - 1) * 79 + m + 63 : 0);
    };
    const int mine = key(h);
    int cnt = 0;
    for (int e = lane; e < total; e += 64) {
        const int w = key(e);
        cnt += (w > mine || (w == mine && e < h)) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o);
    if (lane == 0) map[cnt] = h;
}

// ----------------------------------------------------------------------------------
// batched traceback: the reference's greedy arg-max walk (deepblast/nw.py:401-444,
// sw.py:328-371).  Integer work, bit-identical to the host version in deepblast_amd/_dp.py::traceback,
// including Python's negative-index wrap when exactly one of (i, j) is 0; a walk that leaves the matrix
// (the reference raises IndexError) sets count = -1.
//
// One wavefront per pair.  The walk is a chain of <= N+M dependent steps, each reading three neighbours of the
// current cell; one lane per pair with three global loads per step (round 1) paid a full memory latency per step --
// 0.9 ms at 256 x 512 x 512, more than twice the two sweeps that produce E.  Here the wave keeps the 32 x 32 window
// of E whose bottom-right corner is the current cell in LDS (16 coalesced loads per lane, all in flight together);
// the walk only moves up and left, so it stays inside for 31 ... 62 steps, each three LDS broadcasts and a few scalar
// compares, before the window is re-centred.  Steps are collected one per lane and written 64 at a time from the END
// of the pair's buffer backwards (the reference returns the walk reversed; its length is not known in advance), then
// the wave moves them to the front.  The window is filled through python's index wrap, so the top row / left column
// (floor values, reads that wrap to the opposite edge) and already-wrapped walks use the same loop.
// ----------------------------------------------------------------------------------
// RULE 0: the CPU reference's walk (nw.py:401-444: stop when ALL three neighbours are off the matrix, sentinel -1e5,
// python's index wrap).  RULE 1: the walk of the reference's GPU classes (nw_cuda.py:273-317, sw_cuda.py:283-327: stop
// as soon as ANY neighbour is off the matrix -- or holds the sentinel -1e10; no wrap, never an IndexError).
template <int RULE>
__device__ __forceinline__ void traceback_walk(const float *grad, int *states, int *counts, const int *lens, int B, int N, int M, int cap)
{
    constexpr int TW = SDP_TB_WINDOW, TWL = TW == 64 ? 6 : 5;   // window edge (32 or 64 cells)
    __shared__ float tile[TW * TW];
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= B) return;
    int n = N, m = M;
    if (lens) {
        n = __builtin_amdgcn_readfirstlane(lens[2 * b]);
        m = __builtin_amdgcn_readfirstlane(lens[2 * b + 1]);
        n = n < 1 ? 1 : (n > N ? N : n);
        m = m < 1 ? 1 : (m > M ? M : m);
    }
    const float *g = grad + (size_t)b * N * M;
    int *out = states + (size_t)b * cap * 3;
    const float floor_v = RULE ? -1e10f : -100000.f;
    // A walk has at most n + m - 1 steps: every step lowers i or j, a step that lowers only i needs i > 0, and j never
    // goes below 0.  The API passes cap = N + M + 2.
    if (cap < n + m) {
        if (lane == 0) counts[b] = -1;
        return;
    }
    bool bad = false;
    // every value the walk branches on is the same in all 64 lanes, but a load (LDS or global) is a divergent source to
    // the compiler: these keep the control flow scalar
    auto uni = [](float v) -> float { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
    auto all = [](bool c) -> bool { return __builtin_amdgcn_ballot_w64(c) != 0; };   // c is uniform: any lane == all lanes
    // Steps are recorded as their state only, one per lane; a step moves by (state != 2, state != 0), so the positions
    // of a group of 64 follow from the position before the group and two prefix counts (the first record, state 1 at
    // (n-1, m-1), is "a diagonal step from (n, m)").  Groups go to the END of the pair's buffer, last step first.
    int cnt = 0, my_s = 0;
    int base_i = n, base_j = m;   // position before the first step of the current group
    auto flush = [&](int first, int count, int now_i, int now_j) {  // steps first .. first+count-1 -> positions cap-1-step
        const bool mine = lane < count;
        const unsigned long long mi = __builtin_amdgcn_ballot_w64(mine && my_s != 2), mj = __builtin_amdgcn_ballot_w64(mine && my_s != 0);
        const int pi = __builtin_amdgcn_mbcnt_hi((unsigned)(mi >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mi, 0)) + (int)((mi >> lane) & 1);
        const int pj = __builtin_amdgcn_mbcnt_hi((unsigned)(mj >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mj, 0)) + (int)((mj >> lane) & 1);
        if (mine) {
            int *o = out + 3 * (size_t)(cap - 1 - (first + lane));
            o[0] = base_i - pi, o[1] = base_j - pj, o[2] = my_s;
        }
        base_i = now_i, base_j = now_j;
    };
    auto record = [&](int st, int now_i, int now_j) {  // now = the position after this step
        if (lane == (cnt & 63)) my_s = st;
        ++cnt;
        if ((cnt & 63) == 0) flush(cnt - 64, 64, now_i, now_j);
    };

    // The walk in "virtual" coordinates: i and j only decrease and may go below 0, where python's indexing wraps them
    // to the other edge (nw.py:423 reads grad[i-1, j-1] with i = 0 or j = 0) -- at most once (below -n / -m the
    // reference raises IndexError: bad).  left is off the matrix for i <= 0, upper for j <= 0, all three for both.
    // The window is filled through the same wrap, so one loop serves the interior, the edges and the wrapped walk.
    int i = n - 1, j = m - 1;
    record(1, i, j);
    while (true) {
        if (RULE ? (i <= 0 || j <= 0) : (i <= 0 && j <= 0)) break;   // the reference's stop rule (all three / any one off the matrix)
        const int r0 = i - (TW - 1), c0 = j - (TW - 1);
        __syncthreads();  // one wave: orders the LDS reads of the old window before these writes
        {
            float v[TW * TW / 64];
            int vj = c0 + (lane & (TW - 1));
            vj += vj < 0 ? m : 0;
            const bool okj = vj >= 0;
#pragma unroll
            for (int k = 0; k < TW * TW / 64; ++k) {
                int vi = r0 + (lane >> TWL) + (64 / TW) * k;
                vi += vi < 0 ? n : 0;
                // rows / columns below -n / -m are never read (see `bad` below): clamped address, so that all loads
                // are in flight together
                v[k] = __builtin_nontemporal_load(g + (size_t)max(vi, 0) * M + (okj ? vj : 0));
            }
#pragma unroll
            for (int k = 0; k < TW * TW / 64; ++k) tile[((lane >> TWL) + (64 / TW) * k) * TW + (lane & (TW - 1))] = v[k];
        }
        __syncthreads();
        int ti = TW - 1, tj = TW - 1;   // the current cell; the walk stays in the window while both are >= 1
        bool stop = false;
        if (TW == 32 && r0 >= 0 && c0 >= 0) {
            // ---- the whole window is inside the matrix: no floor values, no wrap (round 6) ----
            // Which way a cell sends the walk depends on the cell alone, so the choices of all 31 x 31 cells of the window are made at
            // once -- the same three comparisons per cell as in the loop below, sixteen cells per lane -- and kept as two bits per cell
            // (0 / 1 / 2: the step; 3: the sentinel rule says stop) in ONE register: lane c + 32 h holds column c, rows 16 h .. 16 h + 15.
            // A step of the walk is then a v_readlane, a shift and a few scalar instructions -- no LDS round trip and no ballot in
            // the chain of <= N + M dependent steps (it was three broadcast reads and three ballots per step: ~360 cycles).
            const int cc = lane & 31, hh = lane >> 5;
            float colv[17];   // rows 16 hh - 1 .. 16 hh + 15 of column cc (row -1 is never a current cell's: clamped)
#pragma unroll
            for (int k = 0; k < 17; ++k) colv[k] = tile[max(16 * hh - 1 + k, 0) * TW + cc];
            unsigned code = 0;
#pragma unroll
            for (int k = 1; k < 17; ++k) {
                // cell (r, cc), r = 16 hh + k - 1: left = (r - 1, cc), diag = (r - 1, cc - 1), upper = (r, cc - 1); column cc - 1 is the
                // lane below's (column 0 has no current cells: the walk leaves the window at tj = 0)
                const float left = colv[k - 1];
                const float diag = __int_as_float(sdp::dpp_i32<sdp::DPP_WAVE_SHR1>(0, __float_as_int(colv[k - 1])));
                const float upper = __int_as_float(sdp::dpp_i32<sdp::DPP_WAVE_SHR1>(0, __float_as_int(colv[k])));
                const bool c1 = diag > left;
                const float bv1 = c1 ? diag : left;
                const bool c2 = upper > bv1;
                const bool halt = RULE ? (left == floor_v || diag == floor_v || upper == floor_v)
                                       : (left == floor_v && diag == floor_v && upper == floor_v);
                code |= (halt ? 3u : (c2 ? 2u : (c1 ? 1u : 0u))) << (2 * (k - 1));
            }
            // The steps themselves run in segments of at most 32 that end where a group of 64 records is complete: inside a segment
            // the states collect in a scalar (two bits per step) and nothing but the look-up, the move and the loop test is in the
            // chain; the lanes take their records -- and a complete group leaves -- between segments.
            while (true) {
                const int cnt0 = cnt;
                const int room = 64 - (cnt0 & 63), lim = room < 32 ? room : 32;
                unsigned long long acc = 0;
                int k = 0;
                bool halted = false;
                while (ti >= 1 && tj >= 1 && k < lim) {
                    const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)code, tj + 32 * (ti >> 4));
                    const int st = (int)((w >> (2 * (ti & 15))) & 3u);
                    if (st == 3) {
                        halted = true;
                        break;
                    }
                    ti -= st == 2 ? 0 : 1;
                    tj -= st == 0 ? 0 : 1;
                    acc = (acc << 2) | (unsigned long long)st;
                    ++k;
                }
                // step t of the segment (0 = first) sits in bits 2 (k - 1 - t) of acc; its record belongs to lane (cnt0 + t) mod 64
                const int t = (lane - cnt0) & 63;
                if (t < k) my_s = (int)((acc >> (2 * (k - 1 - t))) & 3ull);
                cnt = cnt0 + k;
                if (k > 0 && (cnt & 63) == 0) flush(cnt - 64, 64, r0 + ti, c0 + tj);
                if (halted) {
                    stop = true;
                    break;
                }
                if (!(ti >= 1 && tj >= 1)) break;
            }
        } else if (r0 >= 0 && c0 >= 0) {
            // ---- the same for the other window size: three LDS broadcasts and the comparisons per step ----
            while (ti >= 1 && tj >= 1) {
                const float *p = tile + ti * TW + tj;
                const float left = p[-TW], diag = p[-TW - 1], upper = p[-1];
                const bool c1 = all(diag > left);
                const float bv1 = c1 ? diag : left;
                const bool c2 = all(upper > bv1);
                const float bv = c2 ? upper : bv1;
                if constexpr (RULE) {
                    if (all(left == floor_v || diag == floor_v || upper == floor_v)) {   // a stored value equal to the sentinel stops the walk too
                        stop = true;
                        break;
                    }
                } else if (all(bv == floor_v)) {   // only then can all three be the floor value
                    if (all(left == floor_v && diag == floor_v && upper == floor_v)) {
                        stop = true;
                        break;
                    }
                }
                ti -= c2 ? 0 : 1;
                tj -= (c1 || c2) ? 1 : 0;
                record(c2 ? 2 : (c1 ? 1 : 0), r0 + ti, c0 + tj);
            }
        } else {
            while (ti >= 1 && tj >= 1) {
                const int vi = r0 + ti, vj = c0 + tj;
                const bool fl = vi <= 0, fu = vj <= 0;
                if (RULE ? (fl || fu) : (fl && fu)) {
                    stop = true;
                    break;
                }
                if (vi - 1 < -n || vj - 1 < -m) {   // the diagonal read would wrap twice: IndexError in the reference
                    bad = true;
                    break;
                }
                const float *p = tile + ti * TW + tj;
                const float t0 = p[-TW], t1 = p[-TW - 1], t2 = p[-1];
                const float left = fl ? floor_v : uni(t0), diag = uni(t1), upper = fu ? floor_v : uni(t2);
                if (RULE ? (left == floor_v || diag == floor_v || upper == floor_v) : (left == floor_v && diag == floor_v && upper == floor_v)) {
                    stop = true;
                    break;
                }
                int best = 0;
                float bv = left;
                if (diag > bv) best = 1, bv = diag;
                if (upper > bv) best = 2, bv = upper;
                ti -= best == 2 ? 0 : 1;
                tj -= best == 0 ? 0 : 1;
                record(best, r0 + ti, c0 + tj);
            }
        }
        i = r0 + ti, j = c0 + tj;
        if (stop || bad) break;
    }
    while (!bad && i > 0) {
        i -= 1;
        record(0, i, j);
    }
    while (!bad && j > 0) {
        j -= 1;
        record(2, i, j);
    }
    if (bad) {
        if (lane == 0) counts[b] = -1;
        return;
    }
    flush(cnt & ~63, cnt & 63, i, j);
    // the wave reads back what its own lanes stored: workgroup scope is enough (an agent-scope fence writes back and
    // invalidates the L2 on this chip -- tens of microseconds each)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    // steps sit reversed at out[cap-cnt .. cap): move them to the front (ascending groups of 64 never write where a
    // later group still has to read: the source is always at or above the destination)
    const int shift = cap - cnt;
    if (shift > 0) {
        for (int k0 = 0; k0 < cnt; k0 += 64) {
            const int k = k0 + lane;
            int v0 = 0, v1 = 0, v2 = 0;
            if (k < cnt) {
                const int *src = out + 3 * (size_t)(shift + k);
                v0 = __builtin_nontemporal_load(src), v1 = __builtin_nontemporal_load(src + 1), v2 = __builtin_nontemporal_load(src + 2);
            }
            __syncthreads();
            if (k < cnt) {
                int *dst = out + 3 * (size_t)k;
                dst[0] = v0, dst[1] = v1, dst[2] = v2;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        }
    }
    if (lane == 0) counts[b] = cnt;
}

extern "C" __global__ void __launch_bounds__(64) sdp_traceback_kernel(const float *grad, int *states, int *counts,
                                                                      const int *lens, int B, int N, int M, int cap)
{
    traceback_walk<0>(grad, states, counts, lens, B, N, M, cap);
}
extern "C" __global__ void __launch_bounds__(64) sdp_traceback_cuda_kernel(const float *grad, int *states, int *counts,
                                                                           const int *lens, int B, int N, int M, int cap)
{
    traceback_walk<1>(grad, states, counts, lens, B, N, M, cap);
}

// ----------------------------------------------------------------------------------
// masked alignment losses: the reference evaluates its losses with a Python loop over the
// batch -- slice [:x_len, :y_len], masked_select by G, reduce (deepblast/losses.py:9-48, 51-79, 82-118).
// Here one launch reduces every pair (one workgroup per pair, float64 accumulation, deterministic
// order), and one launch writes the gradient w.r.t. the predicted matrix.
//   kind 0 MatrixCrossEntropy : acc = sum_G [ Yt log p + (1-Yt) log(1-p) ],  p = clamp(Yp, 3e-8, 1-3e-8)
//   kind 1 SoftPathLoss       : acc = sum_G (P * Yp)^2
//   kind 2 SoftAlignmentLoss  : acc = sum_G (Yt - Yp)^2
// HBM-bound elementwise work: 12 B read per cell in the forward, 12 B read + 4 B written in the backward.
// ----------------------------------------------------------------------------------
// One workgroup per pair; a thread takes four consecutive columns of a row per iteration (one 16-byte load per
// tensor when the rows are 16-byte aligned, i.e. M a multiple of 4), so the three tensors stream at full width and
// there is one index division per four cells.  Per-thread float64 partial sums, fixed reduction order: deterministic.
extern "C" __global__ void __launch_bounds__(1024) sdp_loss_fwd_kernel(const float *ref, const float *pred, const float *G,
                                                                       const int *lens, double *acc, int *cnt, int N, int M,
                                                                       int kind)
{
    __shared__ double s_acc[16];
    __shared__ int s_cnt[16];
    const int b = blockIdx.x;
    int n = N, m = M;
    if (lens) {
        n = min(max(lens[2 * b], 0), N);
        m = min(max(lens[2 * b + 1], 0), M);
    }
    const size_t base = (size_t)b * N * M;
    double a = 0.0;
    int c = 0;
    const int q4 = (m + 3) >> 2;            // groups of four columns per row
    const int total = n * q4;
    const bool vec = (M & 3) == 0;
    for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
        const int i = idx / q4, j = (idx - i * q4) << 2;
        const size_t o = base + (size_t)i * M + j;
        float g[4], r[4], y[4];
        if (vec && j + 4 <= m) {
            const float4 g4 = *reinterpret_cast<const float4 *>(G + o), r4 = *reinterpret_cast<const float4 *>(ref + o),
                         y4 = *reinterpret_cast<const float4 *>(pred + o);
            g[0] = g4.x, g[1] = g4.y, g[2] = g4.z, g[3] = g4.w;
            r[0] = r4.x, r[1] = r4.y, r[2] = r4.z, r[3] = r4.w;
            y[0] = y4.x, y[1] = y4.y, y[2] = y4.z, y[3] = y4.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool in = j + e < m;
                g[e] = in ? G[o + e] : 0.f;
                r[e] = in ? ref[o + e] : 0.f;
                y[e] = in ? pred[o + e] : 0.5f;
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (g[e] != 0.f) {
                a += (double)sdp::loss_term(r[e], y[e], kind);
                ++c;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off);
        c += __shfl_down(c, off);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_acc[w] = a, s_cnt[w] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0;
        int tc = 0;
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) ta += s_acc[k], tc += s_cnt[k];
        acc[b] = ta;
        cnt[b] = tc;
    }
}

// grid (x, B): the workgroups of a pair stride over groups of four columns of the FULL padded matrix (grad is written
// in full: zero outside the pair's block and where G is 0)
extern "C" __global__ void __launch_bounds__(256) sdp_loss_bwd_kernel(const float *ref, const float *pred, const float *G,
                                                                      const int *lens, const float *scale, float *grad, int N,
                                                                      int M, int kind)
{
    const int b = blockIdx.y;
    int n = N, m = M;
    if (lens) {
        n = min(max(lens[2 * b], 0), N);
        m = min(max(lens[2 * b + 1], 0), M);
    }
    const float sc = scale[b];
    const size_t base = (size_t)b * N * M;
    const int q4 = (M + 3) >> 2;
    const int total = N * q4;
    const bool vec = (M & 3) == 0;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int i = idx / q4, j = (idx - i * q4) << 2;
        const size_t o = base + (size_t)i * M + j;
        float out[4] = {0.f, 0.f, 0.f, 0.f};
        if (i < n && j < m) {
            float g[4], r[4], y[4];
            if (vec) {   // j + 4 <= M: the group lies inside the row (cells at or beyond m are masked below)
                const float4 g4 = *reinterpret_cast<const float4 *>(G + o), r4 = *reinterpret_cast<const float4 *>(ref + o),
                             y4 = *reinterpret_cast<const float4 *>(pred + o);
                g[0] = g4.x, g[1] = g4.y, g[2] = g4.z, g[3] = g4.w;
                r[0] = r4.x, r[1] = r4.y, r[2] = r4.z, r[3] = r4.w;
                y[0] = y4.x, y[1] = y4.y, y[2] = y4.z, y[3] = y4.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool in = j + e < m;
                    g[e] = in ? G[o + e] : 0.f;
                    r[e] = in ? ref[o + e] : 0.f;
                    y[e] = in ? pred[o + e] : 0.5f;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (j + e < m && g[e] != 0.f) out[e] = sdp::loss_dterm(r[e], y[e], sc, kind);
        }
        if (vec) {
            *reinterpret_cast<float4 *>(grad + o) = make_float4(out[0], out[1], out[2], out[3]);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (j + e < M) grad[o + e] = out[e];
        }
    }
}

// ----------------------------------------------------------------------------------
// device self-test of the cross-lane semantics the sweep relies on
// ----------------------------------------------------------------------------------
extern "C" __global__ void sdp_selftest_kernel(int *out)
{
    const int lane = threadIdx.x;
    const double v = 100.0 + lane;
    const double shr = sdp::dpp_f64<sdp::DPP_WAVE_SHR1>(-1.0, v);
    const double shl = sdp::dpp_f64<sdp::DPP_WAVE_SHL1>(-2.0, v);
    const double rol = sdp::dpp_f64<sdp::DPP_WAVE_ROL1>(-3.0, v);
    const double ror = sdp::dpp_f64<sdp::DPP_WAVE_ROR1>(-4.0, v);
    int bad = 0;
    bad |= (shr != (lane == 0 ? -1.0 : 100.0 + lane - 1)) ? 1 : 0;
    bad |= (shl != (lane == 63 ? -2.0 : 100.0 + lane + 1)) ? 2 : 0;
    bad |= (rol != 100.0 + ((lane + 1) & 63)) ? 4 : 0;
    bad |= (ror != 100.0 + ((lane + 63) & 63)) ? 8 : 0;
    // buffer addressing: out-of-range load returns 0, out-of-range store is dropped
    __amdgpu_buffer_rsrc_t r = sdp::make_rsrc(out + 64, 64 * 4);
    const unsigned oob = __builtin_amdgcn_raw_buffer_load_b32(r, sdp::OOB, 0, 0);
    const unsigned neg = __builtin_amdgcn_raw_buffer_load_b32(r, (unsigned)(-4 * (lane + 1)), 0, 0);
    const unsigned past = __builtin_amdgcn_raw_buffer_load_b32(r, 64 * 4 + lane * 4, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(0xdeadu, r, sdp::OOB, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(0xdeadu, r, 64 * 4 + lane * 4, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b32(1000u + lane, r, lane * 4, 0, 0);
    bad |= (oob != 0u) ? 16 : 0;
    bad |= (neg != 0u) ? 32 : 0;
    bad |= (past != 0u) ? 64 : 0;
    out[lane] = bad;
    // informational (sdp_probe): is the scalar offset part of the range check?  Read the word right
    // after the buffer through soffset; 0 = checked (out of range), 7777 = not checked.
    const unsigned via_s = __builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, 64 * 4, 0);
    out[192 + lane] = (int)via_s;
}
#endif  // SDP_IN_GROUP(0)
