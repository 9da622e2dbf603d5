// sdp_kernels.h -- shared between the kernels (sdp_kernels.hip) and the C-ABI host
// side (sdp_api.hip).  Not part of the public interface (that is include/sdp.h).
#ifndef SDP_KERNELS_H_
#define SDP_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sdp.h"

// chunk length K per pass (steps per staged chunk; must divide 64).  K is also the prefetch
// distance of the skewed state rows, in steps.
#ifndef SDP_K_FWD
#define SDP_K_FWD 32
#endif
#ifndef SDP_K_BWD
#define SDP_K_BWD 32
#endif
#ifndef SDP_K_AFWD
#define SDP_K_AFWD 16
#endif
#ifndef SDP_K_ABWD
#define SDP_K_ABWD 16
#endif

// Largest workgroup (in waves) each kernel is compiled for; the VGPR budget per wave is
// 512 / (waves per SIMD), so 8 waves leave 256 registers, 4 waves the full 512.
#ifndef SDP_MAXW_FWD
#define SDP_MAXW_FWD 4
#endif
// Second build of the forward sweep (and the one that writes the exact state) for batches that do not fill
// the GPU: shorter chunks and up to 8 waves shorten the strip pipeline of a single pair.
#ifndef SDP_K_FWD_LAT
#define SDP_K_FWD_LAT 16
#endif
#ifndef SDP_MAXW_FWD_LAT
#define SDP_MAXW_FWD_LAT 8
#endif
#ifndef SDP_MAXW_BWD
#define SDP_MAXW_BWD 4
#endif
// Round 5: the packed-state backward build (K = 32) needs 246 registers since its rare flush paths stopped being hoisted into
// the strip set-up, so EIGHT of its waves fit a CU without a spill -- and for batches that do not fill the GPU they beat the
// K = 16 latency build by a fifth (64 x 512^2: 125 -> 101 us).  The exact-state and parts builds (298+ registers) stay at 4.
#ifndef SDP_MAXW_BWD_Q
#define SDP_MAXW_BWD_Q 8
#endif
// Second build of the backward sweep for batches that do not fill the GPU: shorter chunks and up to 8
// waves shorten the strip pipeline (the kernel is then bound by per-pair latency, not by HBM).
#ifndef SDP_K_BWD_LAT
#define SDP_K_BWD_LAT 16
#endif
#ifndef SDP_MAXW_BWD_LAT
#define SDP_MAXW_BWD_LAT 8
#endif
#ifndef SDP_MAXW_AFWD
#define SDP_MAXW_AFWD 8
#endif
#ifndef SDP_MAXW_ABWD
#define SDP_MAXW_ABWD 4
#endif
#ifndef SDP_DEFAULT_WAVES
#define SDP_DEFAULT_WAVES 4
#endif

namespace sdp {

enum { PASS_FWD = 0, PASS_BWD = 1, PASS_AFWD = 2, PASS_ABWD = 3 };

constexpr int max_waves(int pass)
{
    return pass == PASS_FWD ? SDP_MAXW_FWD : (pass == PASS_BWD ? SDP_MAXW_BWD : (pass == PASS_AFWD ? SDP_MAXW_AFWD : SDP_MAXW_ABWD));
}

// Thin long pairs (fewer than THIN_LO rows or columns, more than THIN_HI of the other) take the exact state: the packed weights'
// rounding does not average out over few paths (sdp_api.hip: exact_for).  thin_pair is evaluated per pair by the kernels of a routed
// launch (Params::route); padded shapes with min(N, M) < THIN_FITS and max > THIN_HI take the exact state as a whole.
constexpr int THIN_LO = 32, THIN_HI = 512, THIN_FITS = 66;
__host__ __device__ constexpr bool thin_pair(int n, int m) { return (n < m ? n : m) < THIN_LO && (n < m ? m : n) > THIN_HI; }
constexpr int MAX_COLS = 2048;     // boundary rows live in LDS (4 x MAX_COLS x 8 B = 64 KiB)
// how a pass represents the values that flow from cell to cell (sdp_kernels.hip, "Carry kinds")
enum { CK_F64 = 0, CK_F32 = 1, CK_EXP = 2 };
// bytes of one slot of a boundary row in LDS: 4 where the values are single floats (the fp32 backward sweep), else 8
__host__ __device__ constexpr int boundary_slot_bytes(int pass) { return pass == PASS_BWD ? 4 : 8; }
constexpr int FRAME_CAP = 136;     // frame words per boundary row: >= 16-step blocks of a strip at MAX_COLS (+ a chunk)
constexpr int PROG_STRIDE = 4096;  // > MAX_COLS: progress words are (use index)*PROG_STRIDE + columns

struct Params {
    const float *sin0;   // staged (row-major) input plane 0: theta | Ztheta | E
    const float *sin1;   // staged input plane 1: A | ZA (may be null = zeros)
    const float *sin2;   // staged input plane 2: G (fused loss seed of the adjoint forward: planes ref, pred, G)
    int loss_kind;       // fused loss seed: SDP_LOSS_*
    float *sout;         // staged (row-major) output: E | Ed
    const uint32_t *qin; // skewed state in: Q, packed (backward sweep) or float2 (adjoint sweeps)
    const float2 *din;   // skewed state in: Qd
    void *dout;          // skewed state out: Q (packed | float2) | Qd (float2)
    const float *vin;    // Et
    int vin_bcast;       // backward sweep: Et is ONE float that applies to every pair (SDP_ET_BROADCAST)
    float *vout;         // Vt | Vtd
    const int32_t *lens; // (B,2) or null
    int B, N, M;
    int nstrips_max;     // ceil(N/64): strips per pair in the state layout
    int tpad;            // state rows (steps) per strip: roundup(M+63, 64)
    size_t st_ps, st2_ps;      // state layout, bytes: stride between (pair, strip) streams -- packed Q | float2 states
    unsigned st_us, st2_us;    // ... and between consecutive 32-step units of one stream
    int mcap;            // doubles per boundary row in LDS
    int stage_off;       // byte offset of the per-wave staging area in LDS
    int variant;
    int route;           // 0: every pair; 1: thin long pairs (thin_pair) are skipped; 2: ONLY they are swept (sdp_api.hip: exact_for)
    int flags;           // bit 0: run every chunk (SDP_NO_ZERO_SKIP), bit 1: no zero fill outside the pairs' blocks (SDP_NO_FILL)
    int dbg;             // experiments build only (sdp_set_debug): bit0 inputs, bit1 outputs, bit2 state: all pairs alias
                         // pair 0; bit3: strips never publish their progress (exercises the hand-off time-out)
    const int *order;    // launch order (pair per workgroup) or null = identity; lives in the tail of the Q state buffer
    int *status;         // host-visible status words of the device: [0] hand-off time-outs, [1..3] first (pair, strip, chunk | pass << 24)
    unsigned long long *trace;   // experiments build only (sdp_set_trace): cycle stamps of the forward sweep's blocks, or null
    int parts;           // strips per workgroup when a pair is spread over several (0 = one workgroup per pair)
    int nparts_max;      // workgroups per pair launched: ceil(ceil(N / 64) / parts)
    unsigned long long *xb;   // bridge between parts: per (pair, boundary between two parts) xb_row granules of 8 bytes
    int xb_row;          // granules per bridged boundary row
    const int *wg_map;   // parts with per-pair lengths: workgroup -> pair * nparts_max + part (sdp_parts_map_kernel), or null
};

// granules per bridged boundary row: one per column, shifted by 63 in the forward sweep (a block of 16 published values
// starts at column 16 j - 63), rounded up to whole 128-byte lines, plus one chunk of slack
__host__ __device__ inline int xb_frame_base(int M) { return (M + 63 + 63) / 64 * 64 + 64; }                 // column granules (+ slack for whole units)
__host__ __device__ inline int xb_row_granules(int M) { return xb_frame_base(M) + (M + 63 + 63) / 64 * 4 + 4; }   // + one frame granule per 16-step block
constexpr unsigned XB_INVALID = 0x7f7f7f7fu;   // tag (high word) of a granule that has not been written: the memset pattern

// per-wave LDS staging (floats): input planes are rings [64][2K], the output ring is [64][2K+1] -- [64][65] in the adjoint backward
// sweep, whose K = 16 builds flush 32-column blocks (sdp_kernels.hip, FLUSH2 / KF), behind a pad of four floats (one is written)
__host__ __device__ constexpr int stage_out_pitch(int pass, int K) { return pass == PASS_ABWD ? 65 : 2 * K + 1; }
__host__ __device__ constexpr int stage_out_pad(int pass) { return pass == PASS_ABWD ? 4 : 0; }
__host__ __device__ constexpr int stage_floats(int pass, int K, int nin_override = 0)
{
    const int nin = nin_override > 0 ? nin_override : ((pass == PASS_FWD || pass == PASS_AFWD) ? 2 : (pass == PASS_ABWD ? 1 : 0));
    const int nout = (pass == PASS_BWD || pass == PASS_ABWD) ? 1 : 0;
    return nin * 64 * (2 * K) + nout * (64 * stage_out_pitch(pass, K) + stage_out_pad(pass));
}

// tail of both state buffers: room for the launch order of a variable-length batch (B ints, 256-byte granules)
__host__ __device__ inline size_t state_order_bytes(int B) { return ((size_t)B * 4 + 255) / 256 * 256; }

// State layout: the state of a (pair, strip) is a sequence of units of 32 steps (packed Q: 10240 B, float2: 16384 B);
// unit u of (pair b, strip s) starts at (b * nstrips + s) * ps + u * us.  See "Skewed state addressing" in sdp_kernels.hip.
// Packed Q: two 20-bit fields per cell, 5 bytes; a 16-step block of a lane is 20 dwords, kept as five rows of 1024 B (row j:
// dwords 4j .. 4j+3 of every lane) -- every wave access is one dwordx4 over contiguous lines; a unit of 32 steps is two blocks,
// ten rows.  (Rounds 1-3: 24-bit fields, 6 bytes; an 18-bit form was built in round 5 and not adopted: sdp_kernels.hip.)
constexpr int STATE_UNIT_STEPS = 32;
constexpr unsigned STATEQ_UNIT_BYTES = 10 * 1024, STATE2_UNIT_BYTES = 32 * 512;
// (Sharing the ramp rows of neighbouring strips -- no skew padding -- was implemented in round 2 for both formats, measured
// slower (partial-line writes) and removed in round 3; see DESIGN.md.)
__host__ __device__ inline size_t state_rows2(int N, int M) { return (size_t)((N + 63) / 64) * ((M + 63 + 63) / 64 * 64); }

// state geometry (shared by host and device)
__host__ __device__ inline int state_nstrips(int N) { return (N + 63) / 64; }
__host__ __device__ inline int state_tpad(int M) { return (M + 63 + 63) / 64 * 64; }

#ifndef SDP_SC_BK
#define SDP_SC_BK 16
#endif
constexpr int SCORES_LDS_BYTES = 2 * 2 * 128 * (SDP_SC_BK + 4) * 4;  // sdp_scores_kernel: [buffer][operand][128 rows][BK + 4 floats]
constexpr int SCORES_X6_LDS_BYTES = 2 * 2 * 3 * 128 * 32;            // sdp_scores_x6_kernel: [buffer][operand][piece][128 rows][32 bytes]
constexpr int SCORES_X6W_LDS_BYTES = 2 * 2 * 3 * 256 * 32;           // sdp_scores_x6w_kernel: the same with 256 rows per operand

}  // namespace sdp

extern "C" {
__global__ void sdp_fwd_kernel(const sdp::Params p);
__global__ void sdp_fwd_lat_kernel(const sdp::Params p);
__global__ void sdp_fwd_x_kernel(const sdp::Params p);
__global__ void sdp_fwd_x_tp_kernel(const sdp::Params p);
__global__ void sdp_fwd_c_kernel(const sdp::Params p);
__global__ void sdp_fwd_x_tp_c_kernel(const sdp::Params p);
__global__ void sdp_fwd_lat_c_kernel(const sdp::Params p);
__global__ void sdp_fwd_x_c_kernel(const sdp::Params p);
__global__ void sdp_bwd_kernel(const sdp::Params p);
__global__ void sdp_bwd_pipe_kernel(const sdp::Params p);
__global__ void sdp_bwd_lat_kernel(const sdp::Params p);
__global__ void sdp_bwd_x_kernel(const sdp::Params p);
__global__ void sdp_bwd_x_lat_kernel(const sdp::Params p);
__global__ void sdp_adj_fwd_kernel(const sdp::Params p);
__global__ void sdp_adj_fwd_loss_kernel(const sdp::Params p);
__global__ void sdp_adj_bwd_kernel(const sdp::Params p);
__global__ void sdp_fwd_g_kernel(const sdp::Params p);
__global__ void sdp_fwd_x_tp_g_kernel(const sdp::Params p);
__global__ void sdp_fwd_p_kernel(const sdp::Params p);
__global__ void sdp_fwd_x_tp_p_kernel(const sdp::Params p);
__global__ void sdp_bwd_p_kernel(const sdp::Params p);
__global__ void sdp_bwd_x_p_kernel(const sdp::Params p);
__global__ void sdp_fwd_pg_kernel(const sdp::Params p);
__global__ void sdp_fwd_x_tp_pg_kernel(const sdp::Params p);
__global__ void sdp_bwd_pg_kernel(const sdp::Params p);
__global__ void sdp_bwd_x_pg_kernel(const sdp::Params p);
__global__ void sdp_bwd_g_kernel(const sdp::Params p);
__global__ void sdp_bwd_lat_g_kernel(const sdp::Params p);
__global__ void sdp_bwd_x_g_kernel(const sdp::Params p);
__global__ void sdp_bwd_x_lat_g_kernel(const sdp::Params p);
__global__ void sdp_adj_bwd_g_kernel(const sdp::Params p);
__global__ void sdp_ref_fwd_kernel(const float *theta, const float *A, float *Q, float *Vt, const int *lens, int N, int M, int sw);
__global__ void sdp_ref_bwd_kernel(const float *Et, const float *Q, float *E, const int *lens, int N, int M, int sw, int et_bcast);
__global__ void sdp_ref_adj_fwd_kernel(const float *Q, const float *Ztheta, const float *ZA, float *Vtd, float *Qd, const int *lens, int N, int M);
__global__ void sdp_ref_adj_bwd_kernel(const float *E, const float *Q, const float *Qd, float *Ed, const int *lens, int N, int M);
// ... and their float64-storage instantiations (sdp_*_f64)
__global__ void sdp_f64_fwd_kernel(const double *theta, const double *A, double *Q, double *Vt, const int *lens, int N, int M, int sw);
__global__ void sdp_f64_bwd_kernel(const double *Et, const double *Q, double *E, const int *lens, int N, int M, int sw, int et_bcast);
__global__ void sdp_f64_adj_fwd_kernel(const double *Q, const double *Ztheta, const double *ZA, double *Vtd, double *Qd, const int *lens, int N, int M);
__global__ void sdp_f64_adj_bwd_kernel(const double *E, const double *Q, const double *Qd, double *Ed, const int *lens, int N, int M);
__global__ void sdp_selftest_kernel(int *out);
__global__ void sdp_loss_fwd_kernel(const float *ref, const float *pred, const float *G, const int *lens, double *acc, int *cnt, int N, int M, int kind);
__global__ void sdp_loss_bwd_kernel(const float *ref, const float *pred, const float *G, const int *lens, const float *scale, float *grad, int N, int M, int kind);
__global__ void sdp_scores_kernel(const float *zx, const float *zy, const float *gx, const float *gy, float *theta, float *A, int B, int N,
                                  int M, int D);
__global__ void sdp_scores_x6_kernel(const float *zx, const float *zy, const float *gx, const float *gy, float *theta, float *A, int B, int N,
                                     int M, int D);
__global__ void sdp_scores_x6s_kernel(const float *zx, const float *zy, const float *gx, const float *gy, float *theta, float *A, int B, int N,
                                     int M, int D);
__global__ void sdp_scores_x6w_kernel(const float *zx, const float *zy, const float *gx, const float *gy, float *theta, float *A, int B, int N,
                                      int M, int D);
__global__ void sdp_scores_bwd_x_kernel(const float *ds0, const float *ds1, const float *y0, const float *y1, float *c0, float *c1, int B, int N, int M, int D);
__global__ void sdp_scores_bwd_y_kernel(const float *ds0, const float *ds1, const float *x0, const float *x1, float *c0, float *c1, int B, int N, int M, int D);
__global__ void sdp_scores_bwd_yf_kernel(const float *g0, const float *g1, const float *act0, const float *act1, float sg0, float sg1, const float *x0,
                                         const float *x1, float *ds0, float *ds1, float *c0, float *c1, int B, int N, int M, int D);
__global__ void sdp_scores_ds_kernel(const float *g_theta, const float *g_A, const float *theta, const float *A, float *ds_theta, float *ds_A, size_t n4);
__global__ void sdp_order_kernel(const int *lens, int *order, int B, int N, int M);
__global__ void sdp_bridge_reset_kernel(unsigned long long *xb, size_t n8);
__global__ void sdp_parts_map_kernel(const int *lens, int *map, int B, int N, int M, int nparts_max, int strips);
__global__ void sdp_traceback_kernel(const float *grad, int *states, int *counts, const int *lens, int B, int N, int M, int cap);
__global__ void sdp_traceback_cuda_kernel(const float *grad, int *states, int *counts, const int *lens, int B, int N, int M, int cap);
}

#endif  // SDP_KERNELS_H_
