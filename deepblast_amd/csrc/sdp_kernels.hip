// sdp_kernels.hip -- wavefront-skewed soft-DP alignment kernels for gfx950 (MI355X).
//
// Replaces the four Numba-CUDA kernels of the reference (one thread per pair, serial
// O(N*M) loop: deepblast/nw_cuda.py:46-165, sw_cuda.py:46-165) with an anti-diagonal
// sweep designed for CDNA4.  Arithmetic follows the CPU reference deepblast/nw.py
// (float64 carries, float32 storage; A indexed [i-1,j-1], nw.py:56-58 -- NOT the GPU
// reference's A[last, j-1], nw_cuda.py:61-63).
//
// Mapping:
//   * one workgroup per pair, W <= 4 wavefronts (one per SIMD);
//   * the N rows are cut into strips of 64; wave w owns strips w, w+W, ...;
//   * inside a strip lane l owns row i0+l and at step t sits on column t-l, so the
//     64 lanes of a wave always lie on one anti-diagonal.  The two predecessors from
//     row i-1 arrive from lane l-1 through one DPP wave shift of a float64 carry (no
//     LDS, no barrier); the row-i predecessor is the lane's own register;
//   * strip-to-strip hand-off (bottom row of strip s -> lane 0 of strip s+1) goes
//     through a float64 row buffer in LDS, published K columns at a time with a
//     monotonic progress word (no s_barrier anywhere);
//   * row-major tensors (theta, A, Ztheta, ZA, E, Ed) are moved in parallelogram
//     chunks of 64 rows x K steps: coalesced buffer loads of K-element row segments
//     -> registers (prefetched one chunk ahead) -> LDS (pitch K+1, conflict-free for
//     both the row-segment writes and the skewed per-lane reads) -> per-step ds_read;
//     outputs take the mirrored path;
//   * the saved state (reference: Q, (B,N+2,M+2,3) fp32) is private to this library,
//     so it is stored ALREADY SKEWED: state[pair][strip][t][lane] = (qx, qy) as
//     float2 (qm = 1 - qx - qy).  Forward writes and backward reads are then single
//     512-byte fully coalesced wave accesses with no transposition.
//   * the reverse passes run the same (t, lane) -> cell schedule backwards in "push"
//     form: each cell scales its E by its own three weights and hands the products to
//     its predecessors, so every cell's weights are read exactly once, by its owner.
//
// No MFMA: this is a scalar recurrence, bounded by HBM bytes and by the length of the
// dependency chain (N+M-1 steps), not by matrix throughput.
#include <hip/hip_runtime.h>
#include <stdint.h>

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

constexpr unsigned OOB = 0x80000000u;

// Ablation switches for timing experiments (tools/gpu_tune.py builds variants with -DSDP_ABL=mask);
// results are wrong when any bit is set.  bit0: no global stores, bit1: no global loads,
// bit2: no strip hand-off waits.
#ifndef SDP_ABL
#define SDP_ABL 0
#endif
constexpr bool ABL_NOSTORE = (SDP_ABL & 1) != 0;
constexpr bool ABL_NOLOAD = (SDP_ABL & 2) != 0;
constexpr bool ABL_NOSYNC = (SDP_ABL & 4) != 0;
// Progress words live in LDS and are polled by other waves.  They are accessed with explicit DS
// instructions: a volatile access through a generic pointer compiles to flat_load + vmcnt(0),
// which drains every outstanding prefetch at each poll.
__device__ __forceinline__ int lds_load_i32(unsigned addr)
{
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void lds_store_i32(unsigned addr, int v)
{
    asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(v) : "memory");
}

template <class X>
__device__ __forceinline__ void keep(X &x)
{
    asm volatile("" : "+v"(x));
}  // voffset that is out of range for every buffer we build

// ----------------------------------------------------------------------------------
// pass descriptions
// ----------------------------------------------------------------------------------
template <int PASS>
struct Traits;
template <>
struct Traits<PASS_FWD> {  // nw.py:46-62
    static constexpr int SIN = 2, SOUT = 0, DIN = 0, DOUT = 1;
    static constexpr bool REV = false;
};
template <>
struct Traits<PASS_BWD> {  // nw.py:120-135
    static constexpr int SIN = 0, SOUT = 1, DIN = 1, DOUT = 0;
    static constexpr bool REV = true;
};
template <>
struct Traits<PASS_AFWD> {  // nw.py:178-199
    static constexpr int SIN = 2, SOUT = 0, DIN = 1, DOUT = 1;
    static constexpr bool REV = false;
};
template <>
struct Traits<PASS_ABWD> {  // nw.py:251-267
    static constexpr int SIN = 1, SOUT = 1, DIN = 2, DOUT = 0;
    static constexpr bool REV = true;
};

// per-lane recurrence state carried from step to step
struct Carry {
    double a;  // fwd: own V (left predecessor)      | rev: value sent to the lane above (px + pm')
    double b;  // fwd: previous `up` (diag predecessor) | rev: own py (to the cell on the left)
    double c;  // rev: pm of the previous step
};

// ----------------------------------------------------------------------------------
// the sweep
// ----------------------------------------------------------------------------------
template <int PASS, int K, int PFD>
__device__ __forceinline__ void sweep(const Params &p)
{
    using T = Traits<PASS>;
    constexpr bool REV = T::REV;
    constexpr int RPI = 64 / K;    // tensor rows covered by one staged load/store instruction
    constexpr int PITCH = K + 1;   // LDS pitch of a staged chunk (floats)
    constexpr int PLANE = 64 * PITCH;
    constexpr int NSTAGE = T::SIN + T::SOUT;
    constexpr int ND = T::DIN > 0 ? T::DIN : 1;
    constexpr int NS = T::SIN > 0 ? T::SIN : 1;

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = blockDim.x >> 6;
    const int b = blockIdx.x;

    int n = p.N, m = p.M;
    if (p.lens) {
        n = p.lens[2 * b];
        m = p.lens[2 * b + 1];
        n = n < 1 ? 1 : (n > p.N ? p.N : n);
        m = m < 1 ? 1 : (m > p.M ? p.M : m);
    }
    n = __builtin_amdgcn_readfirstlane(n);
    m = __builtin_amdgcn_readfirstlane(m);
    const int nstrips = (n + 63) >> 6;
    const int nchunks = (m + 63 + K - 1) / K;  // steps t in [0, m+63)
    const bool sw = p.variant == SDP_SW;

    // ---- LDS carve: boundary rows (f64), progress words, per-wave staging ----
    const int nslot = W > 1 ? W : 2;
    double *bnd = reinterpret_cast<double *>(smem);
    const unsigned prog = (unsigned)(uintptr_t)(bnd + (size_t)nslot * p.mcap);  // LDS byte address of word 0
    float *stage = reinterpret_cast<float *>(smem + p.stage_off) + (size_t)wave * (NSTAGE > 0 ? NSTAGE : 1) * PLANE;
    float *lds_in = stage;
    float *lds_out = stage + T::SIN * PLANE;

    if (threadIdx.x < (unsigned)nslot) lds_store_i32(prog + 4 * threadIdx.x, 0);
    __syncthreads();
    if (wave >= nstrips) return;

    // ---- per-pair tensor descriptors ----
    const size_t plane_elems = (size_t)p.N * p.M;
    const unsigned plane_bytes = (unsigned)(plane_elems * 4);
    __amdgpu_buffer_rsrc_t rs_in[NS];
    if constexpr (T::SIN > 0) {
        rs_in[0] = make_rsrc(p.sin0 + (size_t)b * plane_elems, plane_bytes);
        if constexpr (T::SIN > 1)
            rs_in[1] = make_rsrc(p.sin1 ? p.sin1 + (size_t)b * plane_elems : p.sin0, p.sin1 ? plane_bytes : 0u);
    }
    __amdgpu_buffer_rsrc_t rs_out = make_rsrc(T::SOUT ? (const void *)(p.sout + (size_t)b * plane_elems) : (const void *)p.vout,
                                              T::SOUT ? plane_bytes : 0u);

    // per-lane constants of the staged-chunk geometry: lane -> (row r_l within RPI, step s_l)
    const int r_l = lane / K, s_l = lane % K;
    const int ld = p.M;
    const int lane_off = (r_l * ld + s_l - r_l) * 4;   // byte offset of this lane's element for k = 0, i0 = t0 = 0
    const int lds_rw = r_l * PITCH + s_l;              // LDS slot written (inputs) / read (outputs) by this lane for k = 0
    const int lds_own = lane * PITCH;                  // LDS row this lane reads (inputs) / writes (outputs) per step

    const float et = (PASS == PASS_BWD) ? p.vin[b] : 0.f;

    for (int sidx = wave; sidx < nstrips; sidx += W) {
        const int s = REV ? nstrips - 1 - sidx : sidx;  // strip handled now
        const int i0 = s << 6;
        const int rows = (n - i0) < 64 ? (n - i0) : 64;
        const bool has_pred = REV ? (s + 1 < nstrips) : (s > 0);   // strip whose boundary we consume
        const bool has_succ = REV ? (s > 0) : (s + 1 < nstrips);   // strip that consumes ours
        const int pidx = sidx - 1;                                  // producer's position in processing order
        const int pslot = has_pred ? pidx % nslot : 0, pbase = has_pred ? (pidx / nslot) * PROG_STRIDE : 0;
        const int oslot = sidx % nslot, obase = (sidx / nslot) * PROG_STRIDE;
        double *bnd_in = bnd + (size_t)pslot * p.mcap;
        double *bnd_out = bnd + (size_t)oslot * p.mcap;

        // step at which this lane meets the terminal cell (n-1, m-1) of the pair; -1 if never
        const int t_final = (s == nstrips - 1 && lane == rows - 1) ? (m - 1 + lane) : -1;

        // skewed state addressing: float2 index of (strip s, step 0, lane)
        const size_t st_base = ((size_t)b * p.nstrips_max + s) * p.tpad * 64 + lane;
        const float2 *din[ND];
        if constexpr (T::DIN > 0) {
            din[0] = p.din0 + st_base;
            if constexpr (T::DIN > 1) din[1] = p.din1 + st_base;
        }
        float2 *dout = T::DOUT ? p.dout + st_base : nullptr;

        Carry cy;
        cy.a = 0.0;
        cy.b = 0.0;
        cy.c = 0.0;
        double bc = 0.0;    // boundary values for the edge lane, rotated one lane per step
        double coll = 0.0;  // shift register collecting the edge lane's outputs
        double vt_keep = 0.0;  // fwd passes: value of the terminal cell, captured when this lane reaches it

        float rs[NS][K];       // staged inputs of the NEXT chunk (registers)
        float2 rd[ND][PFD + 1][K];  // direct state rows: ring of chunks, [0] = current

        auto load_staged = [&](int c) {
            if constexpr (T::SIN > 0) {
                const int ubase = (i0 * ld + c * K) * 4;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const unsigned off = (unsigned)(lane_off + ubase + (k * RPI) * (ld - 1) * 4);
#pragma unroll
                    for (int q = 0; q < T::SIN; ++q) {
                        if constexpr (ABL_NOLOAD) {
                            rs[q][k] = __builtin_bit_cast(float, off & 0x3fffffu) * 1e30f;
                        } else {
                            rs[q][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_in[q], off, 0, 0));
                        }
                    }
                }
            }
        };
        auto write_staged = [&]() {
            if constexpr (T::SIN > 0) {
#pragma unroll
                for (int k = 0; k < K; ++k)
#pragma unroll
                    for (int q = 0; q < T::SIN; ++q)
                        lds_in[q * PLANE + lds_rw + k * RPI * PITCH] = rs[q][k];
            }
        };
        auto load_direct = [&](int c, int slot) {
            if constexpr (T::DIN > 0) {
                if (c >= 0 && c < nchunks) {
#pragma unroll
                    for (int k = 0; k < K; ++k)
#pragma unroll
                        for (int q = 0; q < T::DIN; ++q) {
                            if constexpr (ABL_NOLOAD) {
                                rd[q][slot][k] = make_float2(0.25f + 1e-3f * k, 0.5f - 1e-3f * lane);
                            } else {
                                rd[q][slot][k] = din[q][(size_t)(c * K + k) * 64];
                            }
                        }
                }
            }
        };

        const int c_first = REV ? nchunks - 1 : 0;
        const int dir = REV ? -1 : 1;

        // ---- prologue ----
        if constexpr (T::DIN > 0) {
#pragma unroll
            for (int d = 0; d < PFD; ++d) load_direct(c_first + d * dir, d);
        }
        load_staged(c_first);
        write_staged();

        for (int ci = 0; ci < nchunks; ++ci) {
            const int c = REV ? nchunks - 1 - ci : ci;
            const int t0 = c * K;
            const bool more = ci + 1 < nchunks;

            if (more) load_staged(c + dir);
            if constexpr (T::DIN > 0) load_direct(c + PFD * dir, PFD);

            // ---- boundary fetch: K columns for the edge lane ----
            if (has_pred) {
                // fwd: lane 0 needs cols [t0, t0+K); rev: lane 63 needs cols [t0-63, t0+K-63)
                const int c_lo = REV ? t0 - 63 : t0;
                int need;  // progress value that guarantees those columns are published
                if (REV) {
                    const int lo = c_lo < 0 ? 0 : c_lo;
                    need = (c_lo + K > 0 && c_lo < m) ? m - lo : 0;
                } else {
                    const int hi = c_lo + K < m ? c_lo + K : m;
                    need = (c_lo < m) ? hi : 0;
                }
                if (need > 0) {
                    // bounded spin: a missed hand-off must never hang the device (results would be wrong,
                    // which the parity tests catch); ~0.2 s at the cap
                    for (int spin = 0; !ABL_NOSYNC && spin < (1 << 21); ++spin) {
                        if (__builtin_amdgcn_readfirstlane(lds_load_i32(prog + 4 * pslot)) >= pbase + need) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    // fwd: lane l <- col c_lo + l (l < K) ; rev: lane 63-j <- col c_lo + K-1-j
                    const int col = REV ? c_lo + K - 1 - (63 - lane) : c_lo + lane;
                    const bool mine = REV ? (lane >= 64 - K) : (lane < K);
                    bc = (mine && col >= 0 && col < m) ? bnd_in[col] : 0.0;
                } else {
                    bc = 0.0;
                }
            }

            // ---- staged inputs of this chunk: one burst of LDS reads, off the dependency chain ----
            float in0[K], in1[K];
            if constexpr (T::SIN > 0) {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    in0[k] = lds_in[lds_own + k];
                    if constexpr (T::SIN > 1) in1[k] = lds_in[PLANE + lds_own + k];
                }
            }

            // ---- K steps ----
#pragma unroll
            for (int kk = 0; kk < K; ++kk) {
                const int k = REV ? K - 1 - kk : kk;
                const int t = t0 + k;
                const int col = t - lane;
                const bool inside = (unsigned)col < (unsigned)m;
                const bool dead = sw && (col == 0 || (i0 + lane) == 0);  // SW: padded row 1 / col 1

                if constexpr (PASS == PASS_FWD) {
                    const float th = in0[k];
                    const float ga = in1[k];
                    const double up = dpp_f64<DPP_WAVE_SHR1>(bc, cy.a);
                    bc = dpp_f64<DPP_WAVE_ROL1>(bc, bc);
                    const double diag = cy.b, left = cy.a;
                    const double ad = (double)ga;
                    const double x = ad + up, y = ad + left;
                    const double mx = fmax(fmax(x, diag), y);
                    const float ex = fast_exp((float)(x - mx));
                    const float em = fast_exp((float)(diag - mx));
                    const float ey = fast_exp((float)(y - mx));
                    const float ssum = (ex + em) + ey;
                    const float inv = __builtin_amdgcn_rcpf(ssum);
                    const double v = ((double)th + mx) + (double)fast_log(ssum);
                    {
                        float2 qq = make_float2(ex * inv, ey * inv);
                        if constexpr (ABL_NOSTORE) { keep(qq.x); keep(qq.y); } else dout[(size_t)t * 64] = qq;
                    }
                    cy.b = up;
                    cy.a = (col >= 0 && !dead) ? v : 0.0;
                    coll = dpp_f64<DPP_WAVE_SHL1>(cy.a, coll);
                    vt_keep = (t == t_final) ? cy.a : vt_keep;
                } else if constexpr (PASS == PASS_AFWD) {
                    const float zt = in0[k];
                    const float za = in1[k];
                    float2 q = rd[0][0][k];
                    const double up = dpp_f64<DPP_WAVE_SHR1>(bc, cy.a);
                    bc = dpp_f64<DPP_WAVE_ROL1>(bc, bc);
                    const double diag = cy.b, left = cy.a;
                    const bool live = inside && !dead;
                    const double qx = live ? (double)q.x : 0.0, qy = live ? (double)q.y : 0.0;
                    const double qm = live ? (1.0 - qx) - qy : 0.0;
                    const double zad = (double)za;
                    const double a0 = zad + up, a1 = diag, a2 = zad + left;
                    const double tot = (qx * a0 + qm * a1) + qy * a2;
                    const double vd = (double)zt + tot;
                    {
                        float2 qq = make_float2((float)(qx * (a0 - tot)), (float)(qy * (a2 - tot)));
                        if constexpr (ABL_NOSTORE) { keep(qq.x); keep(qq.y); } else dout[(size_t)t * 64] = qq;
                    }
                    cy.b = up;
                    cy.a = inside ? vd : 0.0;
                    coll = dpp_f64<DPP_WAVE_SHL1>(cy.a, coll);
                    vt_keep = (t == t_final) ? cy.a : vt_keep;
                } else if constexpr (PASS == PASS_BWD) {
                    float2 q = rd[0][0][k];
                    const double in = dpp_f64<DPP_WAVE_SHL1>(bc, cy.a);
                    bc = dpp_f64<DPP_WAVE_ROR1>(bc, bc);
                    const bool live = inside && lane < rows && !dead;
                    double e = in + cy.b;
                    e = (t == t_final) ? (double)et : e;
                    e = live ? e : 0.0;
                    const double qx = live ? (double)q.x : 0.0, qy = live ? (double)q.y : 0.0;
                    const double qm = (1.0 - qx) - qy;
                    const double px = qx * e, pm = qm * e;
                    cy.b = qy * e;
                    cy.a = px + cy.c;
                    cy.c = pm;
                    coll = dpp_f64<DPP_WAVE_SHR1>(cy.a, coll);
                    lds_out[lds_own + k] = (float)e;
                } else {  // PASS_ABWD
                    float2 q = rd[0][0][k];
                    float2 qd = rd[1][0][k];
                    const float ef = in0[k];
                    const double in = dpp_f64<DPP_WAVE_SHL1>(bc, cy.a);
                    bc = dpp_f64<DPP_WAVE_ROR1>(bc, bc);
                    const bool cell = inside && lane < rows;
                    const bool live = cell && !dead;
                    const double ed = cell ? in + cy.b : 0.0;
                    const double e = live ? (double)ef : 0.0;
                    const double qx = live ? (double)q.x : 0.0, qy = live ? (double)q.y : 0.0;
                    const double qm = live ? (1.0 - qx) - qy : 0.0;
                    const double dx = live ? (double)qd.x : 0.0, dy = live ? (double)qd.y : 0.0;
                    const double dm = -(dx + dy);
                    const double gx = dx * e + qx * ed;
                    const double gm = dm * e + qm * ed;
                    cy.b = dy * e + qy * ed;
                    cy.a = gx + cy.c;
                    cy.c = gm;
                    coll = dpp_f64<DPP_WAVE_SHR1>(cy.a, coll);
                    lds_out[lds_own + k] = (float)ed;
                }
            }

            // ---- publish K boundary values for the next strip ----
            if (has_succ) {
                // fwd: lane 63-j holds col (t0+K-1-63) - j ; rev: lane j holds col t0 + j   (j < K newest)
                const int col = REV ? t0 + lane : t0 + K - 1 - 63 - (63 - lane);
                const bool mine = REV ? (lane < K) : (lane >= 64 - K);
                if (mine && col >= 0 && col < m) bnd_out[col] = coll;
                int done;  // columns published so far (fwd: from the left; rev: from the right)
                if (REV) {
                    done = t0 < m ? m - t0 : 0;
                } else {
                    const int hi = t0 + K - 63;
                    done = hi < 0 ? 0 : (hi > m ? m : hi);
                }
                // LDS executes a wave's DS instructions in order, so the data written above is visible to
                // any wave that observes this word (the asm statements also stop compiler reordering)
                if (lane == 0) lds_store_i32(prog + 4 * oslot, obase + done);
            }

            // ---- flush the staged output chunk (row-major, coalesced row segments) ----
            if constexpr (T::SOUT > 0) {
                const int ubase = i0 * ld + t0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const int row = k * RPI + r_l;
                    const int col = t0 + s_l - row;
                    const float val = lds_out[lds_rw + k * RPI * PITCH];
                    const bool ok = (unsigned)col < (unsigned)m && (i0 + row) < n;
                    const unsigned off = ok ? (unsigned)(lane_off + (ubase + k * RPI * (ld - 1)) * 4) : OOB;
                    if constexpr (ABL_NOSTORE) {
                        unsigned vv = __builtin_bit_cast(unsigned, val) ^ off;
                        keep(vv);
                    } else {
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rs_out, off, 0, 0);
                    }
                }
            }

            // ---- rotate prefetch buffers ----
            if (more) write_staged();
            if constexpr (T::DIN > 0) {
#pragma unroll
                for (int d = 0; d < PFD; ++d)
#pragma unroll
                    for (int k = 0; k < K; ++k)
#pragma unroll
                        for (int q = 0; q < T::DIN; ++q) rd[q][d][k] = rd[q][d + 1][k];
            }
        }
        if constexpr (!REV) {
            if (t_final >= 0) p.vout[b] = (float)vt_keep;
        }
    }
}

}  // namespace sdp

// ----------------------------------------------------------------------------------
// kernels (one symbol per pass so that rocprofv3 names them)
// ----------------------------------------------------------------------------------
#define SDP_KERNEL(NAME, PASS, K, PFD)                                                     \
    extern "C" __global__ void __launch_bounds__(256) NAME(const sdp::Params p)           \
    {                                                                                      \
        sdp::sweep<PASS, K, PFD>(p);                                                       \
    }

SDP_KERNEL(sdp_fwd_kernel, sdp::PASS_FWD, SDP_K_FWD, 0)
SDP_KERNEL(sdp_bwd_kernel, sdp::PASS_BWD, SDP_K_BWD, SDP_PFD_BWD)
SDP_KERNEL(sdp_adj_fwd_kernel, sdp::PASS_AFWD, SDP_K_AFWD, SDP_PFD_AFWD)
SDP_KERNEL(sdp_adj_bwd_kernel, sdp::PASS_ABWD, SDP_K_ABWD, SDP_PFD_ABWD)

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
}
