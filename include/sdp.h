/*
 * sdp.h -- C ABI of the MI355X soft-DP alignment engine (libsdp_hip.so).
 *
 * This is the drop-in boundary for DeepBLAST's differentiable alignment operator.
 * Each entry point replaces one Numba-CUDA kernel launch of the reference; the
 * Python side (deepblast_amd/nw.py, sw.py) binds them with ctypes exactly where
 * the reference launches its kernels:
 *
 *   sdp_forward_f32           <- _forward_pass_kernel[tpb,bpg](theta, A, Q, Vt)
 *                                deepblast/nw_cuda.py:74-79,184-187  (sw_cuda.py:74-79)
 *   sdp_backward_f32          <- _backward_pass_kernel[tpb,bpg](Et, Q, E)
 *                                deepblast/nw_cuda.py:98-102,222-226 (sw_cuda.py:98-102)
 *   sdp_adjoint_forward_f32   <- _adjoint_forward_pass_kernel[tpb,bpg](Q, Ztheta, ZA, Vtd, Qd)
 *                                deepblast/nw_cuda.py:134-139,258
 *   sdp_adjoint_backward_f32  <- _adjoint_backward_pass_kernel[tpb,bpg](E, Q, Qd, Ed)
 *                                deepblast/nw_cuda.py:160-165,259
 *   sdp_state_bytes,
 *   sdp_state_d_bytes         <- torch.zeros((B, N+2, M+2, 3)) for Q / Qd
 *                                deepblast/nw_cuda.py:180-182,250-252
 *
 * Conventions
 *   - Every tensor pointer is a DEVICE pointer to a contiguous row-major fp32 array
 *     owned by the caller (PyTorch's caching allocator).  The library never
 *     allocates, frees or retains device memory.
 *   - theta, A, ZA, Ztheta, E, Ed are (B, N, M).  E/Ed are written in full
 *     (the reference's (B,N+2,M+2) zero border is not materialised; its interior
 *     [:,1:-1,1:-1] is exactly this array).
 *   - `state` / `state_d` are opaque buffers of sdp_state_bytes(B,N,M) /
 *     sdp_state_d_bytes(B,N,M) bytes that stand in for the reference's Q / Qd
 *     tensors.  Their layout is private (wavefront-skewed); only
 *     this library reads them.  For the backward sweep Q is kept as two 20-bit
 *     fixed-point weights per cell (absolute error <= 2^-21 = 4.8e-7 per weight,
 *     SDP_PACKED_STATE_BYTES_PER_CELL = 5 bytes per cell, times 1.125 for the skew
 *     padding at M = 512; a weight within 2^-21 of 1 / of 0 decodes to exactly 1 / 0).
 *     The adjoint sweeps (second order) multiply the weights with
 *     directional derivatives of any size and need them at full fp32 precision:
 *     run sdp_forward_f32 with SDP_EXACT_STATE for them (float2 per cell, the
 *     size of Qd); sdp_backward_f32 reads that format too when given the flag.
 *     Problems with N + M > 4096 always use the float2 form (the packed format's
 *     rounding error is carried along an alignment path like a random walk: measured
 *     <= 4e-5 of E at N = M = 2048 on soft and on steep scores, bound 1e-4),
 *     and so do THIN long problems -- fewer than 32 rows or columns with more than
 *     512 of the other, where those errors do not average out over many paths
 *     (2 x 2048, flat scores: 1.0e-4 packed, 4.5e-6 exact).  Round 6: that holds PER
 *     PAIR when `lens` is given -- a thin long pair inside a fat padded batch is
 *     swept by the float2 build in a second launch over the same buffers (its state
 *     lives inside the pair's own record), every other pair by the packed build; padded
 *     shapes with min(N, M) < 66 and max(N, M) > 512 use the float2 form as a whole.
 *     sdp_state_bytes accounts for all of it, and forward and backward apply the
 *     same rule, so callers need not care.
 *   - `lens` is NULL (reference semantics: every pair uses the full padded N x M)
 *     or a DEVICE pointer to B x 2 int32 (n_b, m_b): pair b is aligned over its
 *     top-left n_b x m_b block, terminal cell (n_b, m_b); E/Ed outside the block
 *     are written as zero.
 *   - variant: SDP_NW (deepblast/nw.py) or SDP_SW (deepblast/sw.py: the same
 *     recurrence with padded row 1 / column 1 skipped in forward and backward).
 *   - `device` is the HIP device ordinal, `stream` a hipStream_t (NULL = default
 *     stream).  Calls only enqueue work; they never synchronise.
 *   - Return: 0 ok; negative = SDP_E_* below; positive = hipError_t.  Nothing is
 *     thrown across the boundary.  sdp_last_error_string() is thread-local.
 *   - Re-entrant: calls from several threads / on several streams may overlap.  Process-wide state is limited
 *     to a per-thread error string and one 64-byte block of host-pinned status words per device (created by
 *     sdp_init or on the first launch there, published under a lock and read through atomics) through which a kernel reports a strip hand-off that timed out: such a launch's
 *     results are invalid, and the NEXT call on that device (or sdp_device_status) returns SDP_E_HANDOFF once.
 *     There is no tuning state: a wave-count override travels with the call (SDP_WAVES).
 */
#ifndef SDP_H_
#define SDP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SDP_VERSION 106 /* 0.1.5: per-pair state format under `lens` (thin long pairs routed to the float2 build); 0.1.4: + SDP_NO_ZERO_SKIP, SDP_NO_FILL */

#define SDP_NW 0
#define SDP_SW 1
/* or-ed into `variant` of sdp_forward_f32: `state` (then sdp_state_d_bytes large) receives Q at full fp32
 * precision; or-ed into `variant` of sdp_backward_f32: `state` is such a buffer.  The two adjoint entry points
 * always require a state produced this way.  A caller that knows the second-order sweeps will follow (training:
 * decode() + loss.backward()) runs forward and backward with the flag and shares the one state; a caller that
 * only needs E (inference) uses the default compact state, which the backward sweep reads faster. */
#define SDP_EXACT_STATE 0x100

/* or-ed into `variant` of sdp_backward_f32 / sdp_backward_range_f32: `Et` points at ONE float that applies to every
 * pair -- the cotangent `Vt.sum().backward()` hands over is a broadcast scalar, and expanding it into (B,) floats first
 * would be a kernel launch of its own between the two sweeps. */
#define SDP_ET_BROADCAST 0x200

/* or-ed into `variant` of the four sweeps: the REFERENCE's arithmetic, rounding for rounding (csrc/sdp_ref.hip) --
 * float64 exp / log / division and one rounding of Q to fp32 in the forward sweep (deepblast/nw.py:10-27, 115), all three
 * weights kept; the soft-max Hessian product and the Qd * E products formed in fp32 exactly where numpy forms them
 * (nw.py:30-43, 261-266); everything else float64.  The default sweeps keep those products in float64 and are the more
 * accurate ones; on long saturated alignments (N + M beyond ~2500 with |theta| beyond ~50, or positive gap scores on long
 * thin problems) the reference's own fp32 roundings move Ed by 1-2e-4, and only this mode reproduces that (to ~1e-7).
 * All four sweeps of a problem must use the flag or none: the states are then the reference's, (B, N, M, 3) fp32 each
 * (sdp_state_bytes_v / sdp_state_d_bytes_v).  Unoptimised: milliseconds where the default path takes a fraction of one.
 * Not available for sdp_adjoint_forward_loss_f32. */
#define SDP_REF_ROUNDING 0x400
/* or-ed into `variant` of sdp_backward_f32 / sdp_backward_range_f32 / sdp_adjoint_backward_f32: run EVERY chunk of the sweep.  By default the fp32
 * backward sweep neither runs 32-step chunks that can only produce +0 nor reads their state (E underflows to exactly +0
 * away from the alignment) -- a data-dependent saving.  The results are bit-identical either way; the flag
 * is the control a measurement needs (bench.py reports both). */
#define SDP_NO_ZERO_SKIP 0x800
/* or-ed into `variant` of sdp_backward_f32 / sdp_adjoint_backward_f32 when `lens` is given: do NOT zero E / Ed outside
 * each pair's n_b x m_b block -- those cells keep whatever the buffer held.  For callers that never read them: a loss
 * that masks by the same lengths (deepblast/losses.py:30-40 slices [:x_len, :y_len]), the batched traceback with
 * lengths, a gather of walks.  The zero fill of a padded batch is as many bytes again as the sweep moves
 * (BASELINE configs[2]: 784 MB of zeros next to 724 MB). */
#define SDP_NO_FILL 0x10000
/* bytes per cell of the packed state (the header's statement of the format; tests/test_abi.py holds sdp_state_bytes to it) */
#define SDP_PACKED_STATE_BYTES_PER_CELL 5

#define SDP_E_NULLPTR (-1)  /* a required pointer is NULL */
#define SDP_E_SHAPE (-2)    /* B, N or M non-positive */
#define SDP_E_MAXCOLS (-3)  /* M exceeds sdp_max_cols() (reference: max_cols, nw_cuda.py:11) */
#define SDP_E_VARIANT (-4)  /* variant is neither SDP_NW nor SDP_SW */
#define SDP_E_TOOBIG (-5)   /* a tensor exceeds the 4 GiB-per-plane addressing limit */
#define SDP_E_SELFTEST (-6) /* sdp_selftest found a hardware-semantics mismatch */
#define SDP_E_COMM (-8)     /* RCCL could not be loaded or returned an error (sdp_comm_last_error_string) */
#define SDP_E_HANDOFF (-7)  /* an earlier launch on this device timed out waiting for a strip hand-off: its results are invalid */

/* or-ed into `variant` of the four sweeps: run with w (1..8) wavefronts per pair instead of the automatic choice
 * (clamped to what the kernel build and the problem allow).  Results do not depend on it (bit-identical). */
#define SDP_WAVES(w) (((w) & 0xf) << 12)

int sdp_version(void);

/* Human-readable description of the last non-zero return on this thread. */
const char *sdp_last_error_string(void);

/* Largest M accepted (the reference's GPU path stops at 2047 columns). */
int sdp_max_cols(void);

/* Bytes of the opaque buffers that hold Q (`state`) and Qd (`state_d`) for a (B,N,M) problem; 0 on bad shape. */
size_t sdp_state_bytes(int B, int N, int M);
size_t sdp_state_d_bytes(int B, int N, int M);

/* ... the same for a given `variant` word (SDP_EXACT_STATE, SDP_REF_ROUNDING): what the sweeps called with that word
 * read and write. */
size_t sdp_state_bytes_v(int B, int N, int M, int variant);
size_t sdp_state_d_bytes_v(int B, int N, int M, int variant);

/* Vt[b] = V[n_b, m_b]; state <- softmax weights of every cell. */
int sdp_forward_f32(const float *theta, const float *A, float *state, float *Vt, int B, int N,
                    int M, const int32_t *lens, int variant, int device, void *stream);

/* E = dVt/dtheta * Et  (expected alignment matrix), from the saved state. */
int sdp_backward_f32(const float *Et, const float *state, float *E, int B, int N, int M,
                     const int32_t *lens, int variant, int device, void *stream);

/* The backward sweep over pairs first .. first + count - 1 of the batch only: Et, state and E are the WHOLE batch's
 * buffers (B pairs, no per-pair lengths), E[first .. first + count) is written and nothing else is touched.  Bit-identical
 * to the same rows of one sdp_backward_f32 over the batch.  For callers that hand E on in pieces -- the chunked all-gather
 * of deepblast_amd/distributed.py: the collective on piece k runs under the sweep of piece k + 1.  The library
 * locates a pair's records itself (sdp_state_pair_stride: bytes between the records of consecutive pairs in `state`, for
 * callers that want to know -- NOT sdp_state_bytes(2) - sdp_state_bytes(1), which also counts the per-pair share of the
 * buffer's tail); launches over part of a batch never spread a pair over several workgroups (sdp_plan_parts), whose
 * bridge rows live in that tail.  variant: SDP_NW / SDP_SW, | SDP_EXACT_STATE as for sdp_backward_f32, | SDP_WAVES(w). */
size_t sdp_state_pair_stride(int N, int M, int exact_state);
int sdp_backward_range_f32(const float *Et, const float *state, float *E, int B, int N, int M, int first, int count,
                           int variant, int device, void *stream);

/* Directional derivative through the DP: Vtd (B,), state_d <- Qd.  ZA may be NULL (= zeros).
 * `state` must come from sdp_forward_f32(..., variant | SDP_EXACT_STATE, ...). */
int sdp_adjoint_forward_f32(const float *state, const float *Ztheta, const float *ZA, float *Vtd,
                            float *state_d, int B, int N, int M, const int32_t *lens, int variant,
                            int device, void *stream);

/* Ed = reverse sweep of the derivative (Hessian-vector product w.r.t. theta); `state` as for the adjoint
 * forward sweep (exact). */
int sdp_adjoint_backward_f32(const float *E, const float *state, const float *state_d, float *Ed,
                             int B, int N, int M, const int32_t *lens, int variant, int device,
                             void *stream);

/* float64 tensors.  The reference's CPU classes take whatever dtype they are given -- its own tests run the decoding
 * test, gradcheck and gradgradcheck on float64 tensors (deepblast/tests/test_nw.py:46-90, test_sw.py) -- so the drop-in
 * does too: the four sweeps with float64 storage and float64 arithmetic throughout (the recurrences of nw.py / sw.py
 * as numpy evaluates them on float64 arrays).  Same arguments as the _f32 entry points; theta, A, Vt, Et, E, Ztheta, ZA,
 * Vtd, Ed are float64, `state` and `state_d` are the reference's (B, N, M, 3) weights in float64
 * (sdp_state_bytes_f64 bytes each), `variant` is SDP_NW / SDP_SW (| SDP_ET_BROADCAST for the backward sweep).  One
 * workgroup per pair and a barrier per anti-diagonal (csrc/sdp_ref.hip): a path for tests and small problems --
 * milliseconds where the float32 path takes a fraction of one -- not a second fast path. */
size_t sdp_state_bytes_f64(int B, int N, int M);
int sdp_forward_f64(const double *theta, const double *A, double *state, double *Vt, int B, int N, int M,
                    const int32_t *lens, int variant, int device, void *stream);
int sdp_backward_f64(const double *Et, const double *state, double *E, int B, int N, int M,
                     const int32_t *lens, int variant, int device, void *stream);
int sdp_adjoint_forward_f64(const double *state, const double *Ztheta, const double *ZA, double *Vtd, double *state_d,
                            int B, int N, int M, const int32_t *lens, int variant, int device, void *stream);
int sdp_adjoint_backward_f64(const double *E, const double *state, const double *state_d, double *Ed,
                             int B, int N, int M, const int32_t *lens, int variant, int device, void *stream);

/* The score tensors the DP reads (reference: NeuralAligner.forward / .score, deepblast/alignment.py:122-123, 134-135:
 *   theta = F.softplus(torch.einsum('bid,bjd->bij', zx, zy));  A = F.logsigmoid(torch.einsum('bid,bjd->bij', gx, gy))).
 * zx, gx: (B,N,D); zy, gy: (B,M,D); theta, A: (B,N,M); all fp32, contiguous.  gx, gy and A may be NULL together
 * (theta only).  One launch: batched GEMM on the matrix cores with the activation applied to the accumulators.  Two
 * kernels, chosen per call: D a multiple of 16 and 16-byte aligned embeddings take the bf16 pipe with every fp32 operand
 * cut into three exact bf16 pieces and six piece products per k (the dropped pairs are <= 2^-23 of a product: fp32
 * accuracy, measured error vs a float64 einsum 1.9e-7 at D = 512, the same as torch's fp32 einsum); anything else takes
 * the f32-input MFMA (exact fp32 products and sums).  Accumulation is fp32 in both.  Inputs must be finite: in the
 * three-piece kernel an Inf operand gives NaN (Inf - Inf in the cut) where the f32 kernel and torch give Inf. */
int sdp_scores_f32(const float *zx, const float *zy, const float *gx, const float *gy, float *theta, float *A, int B, int N,
                   int M, int D, int device, void *stream);

/* Backward of sdp_scores_f32: the gradients of the embeddings from the gradients of theta and A (the reference gets them
 * from autograd through its two einsums and activations, alignment.py:122-123).  With dS = g * d act / ds formed from the
 * saved OUTPUTS -- g_theta * (1 - exp(-theta)), g_A * (1 - exp(A)):  dzx[b,i,:] = sum_j dS_theta[b,i,j] zy[b,j,:],
 * dzy[b,j,:] = sum_i dS_theta[b,i,j] zx[b,i,:], and dgx, dgy from (dS_A, gy, gx).  The same three-piece bf16 product as the
 * forward (fp32 accuracy), 256 x 256 tiles, one launch per side for both tensors: the dzy / dgy product forms dS on its way
 * into LDS and leaves a copy in `ws` (sdp_scores_backward_ws_bytes; caller-owned scratch) for the dzx / dgx product.  (g_A, A, gx, gy, dgx, dgy) may be NULL together (theta only), likewise the theta group.
 * Needs M and D multiples of 4 and 16-byte aligned tensors: otherwise SDP_E_SHAPE (the Python layer then uses
 * torch.bmm).  Inputs must be finite. */
size_t sdp_scores_backward_ws_bytes(int B, int N, int M);
int sdp_scores_backward_f32(const float *g_theta, const float *g_A, const float *theta, const float *A, const float *zx,
                            const float *zy, const float *gx, const float *gy, float *ws, float *dzx, float *dzy, float *dgx,
                            float *dgy, int B, int N, int M, int D, int device, void *stream);

/* Batched traceback (reference: Decoder.traceback, deepblast/nw.py:401-444, called once per pair by
 * NeuralAligner.traceback, alignment.py:165-170).  grad is (B,N,M); states receives, per pair, up to
 * sdp_traceback_capacity(N,M) triples (i, j, state) in the reference's order (start of the alignment first),
 * counts[b] the number of triples, or -1 where the reference's walk would raise IndexError.  Rows of `states`
 * past counts[b] are left as they were (scratch). */
int sdp_traceback_capacity(int N, int M);
int sdp_traceback_i32(const float *grad, int32_t *states, int32_t *counts, int B, int N, int M,
                      const int32_t *lens, int device, void *stream);
/* The same with the walk rule chosen: SDP_TRACEBACK_CPU = the CPU classes' walk (deepblast/nw.py:401-444, sw.py:328-371:
 * stop when ALL three neighbours are off the matrix, sentinel -1e5, Python's negative-index wrap at the edges; what
 * sdp_traceback_i32 does), SDP_TRACEBACK_CUDA = the walk of the classes this library replaces (deepblast/nw_cuda.py:
 * 273-317, sw_cuda.py:283-327: stop as soon as ANY neighbour is off the matrix or holds the sentinel -1e10; never
 * wraps, counts[b] is never -1).  The two differ when a walk reaches row 0 or column 0 before the other. */
#define SDP_TRACEBACK_CPU 0
#define SDP_TRACEBACK_CUDA 1
int sdp_traceback_rule_i32(const float *grad, int32_t *states, int32_t *counts, int B, int N, int M,
                           const int32_t *lens, int rule, int device, void *stream);

/* Masked alignment losses (reference: deepblast/losses.py -- MatrixCrossEntropy :9-48, SoftPathLoss :51-79,
 * SoftAlignmentLoss :82-118; evaluated there with a Python loop over the batch, trainer.py:154-171).
 * ref = Ytrue (kinds 0, 2) or the path-distance matrix P (kind 1); pred = predicted alignment matrix;
 * G = mask (non-zero = counted); all (B,N,M) fp32.  Forward: acc[b] = per-pair masked sum (see kernel
 * header), cnt[b] = number of counted cells.  Backward: grad (B,N,M), written in full, = scale[b] times the
 * per-element derivative factor.  The Python layer (deepblast_amd/losses.py) turns acc/cnt into the
 * reference's scalar and supplies scale. */
#define SDP_LOSS_CROSS_ENTROPY 0
#define SDP_LOSS_PATH 1
#define SDP_LOSS_ALIGNMENT 2
int sdp_loss_forward_f32(const float *ref, const float *pred, const float *G, const int32_t *lens, double *acc,
                         int32_t *cnt, int B, int N, int M, int kind, int device, void *stream);
int sdp_loss_backward_f32(const float *ref, const float *pred, const float *G, const int32_t *lens,
                          const float *scale, float *grad, int B, int N, int M, int kind, int device,
                          void *stream);

/* EXPERIMENTAL -- parity-equal to the unfused sequence, but SLOWER than it (B=256, 512 x 512: 2.02 vs 1.62 ms per training
 * step; the seed's divisions sit on the sweep's dependency chain and cost more than the 268 MB tensor they save).  Kept
 * for callers who are short of memory, not of time; deepblast_amd.losses uses the unfused kernels by default.
 * The adjoint forward sweep with the loss's gradient as its seed, formed inside the kernel: Ztheta[b,i,j] =
 * scale[b] * d(term)/d(pred) where G != 0 (and inside the pair's block), 0 elsewhere -- exactly what
 * sdp_loss_backward_f32 would write and sdp_adjoint_forward_f32 would read back, without the (B,N,M) tensor in
 * between (training: decode -> masked loss on the alignment matrix -> backward; reference: losses.py:9-118 applied to
 * NeuralAligner.forward's output, trainer.py:154-171).  pred is the alignment matrix E the loss was evaluated on;
 * ZA is taken as zero.  state as for sdp_adjoint_forward_f32. */
int sdp_adjoint_forward_loss_f32(const float *state, const float *ref, const float *pred, const float *G,
                                 const float *scale, int kind, float *Vtd, float *state_d, int B, int N, int M,
                                 const int32_t *lens, int variant, int device, void *stream);

/* Collecting results across the GPUs of a node for callers without torch.distributed.  The sweeps need no
 * collective; these four wrap the one RCCL all-gather (over xGMI) that gathers Vt -- or E -- from all ranks, one
 * process per GPU.  Rank 0 calls sdp_comm_unique_id and distributes the 128 bytes to the other ranks by its own means;
 * every rank then calls sdp_comm_init (collective) with its device, and sdp_comm_all_gather_f32 enqueues the gather of
 * count_per_rank floats per rank on `stream` (recv holds world * count_per_rank floats, rank order).  RCCL is loaded
 * at run time (a copy already in the process, e.g. PyTorch's, is reused).  Errors: SDP_E_COMM + sdp_comm_last_error_string. */
int sdp_comm_unique_id(void *id128);
int sdp_comm_init(void **comm, const void *id128, int rank, int world, int device);
int sdp_comm_all_gather_f32(void *comm, const float *send, float *recv, size_t count_per_rank, void *stream);
int sdp_comm_destroy(void *comm);
const char *sdp_comm_last_error_string(void);

/* Optional: creates the per-device status words and raises the kernels' dynamic-LDS limit on `device` now instead of
 * inside the first launch there (both are host-side, once per device / per calling thread).  Call it before capturing
 * launches into a hipGraph: an allocation is not allowed inside a capture. */
int sdp_init(int device);

/* Runs a few-microsecond device check of the cross-lane (DPP) and buffer-addressing
 * behaviour the kernels rely on.  Synchronises the device.  0 = ok. */
int sdp_selftest(int device);

/* Status words of `device`: info[0] = strip hand-offs that timed out since the library was loaded, info[1..3] =
 * pair, strip, chunk | pass << 24 of the first one.  Returns SDP_E_HANDOFF if there are time-outs that no call has
 * reported yet, else 0.  Host-side read, no synchronisation: synchronise the stream first to cover its launches. */
int sdp_device_status(int device, int32_t info[4]);

/* Diagnostic: what a launch of pass (0 fwd, 1 bwd, 2 adj-fwd, 3 adj-bwd) would use on a device with `cus` compute
 * units -- kernel build (0 fwd throughput, 1 bwd throughput, 2 adj-fwd, 3 adj-bwd, 4 bwd latency, 5 fwd exact
 * state (latency), 6 fwd latency, 7 / 8 bwd reading the exact state (throughput / latency), 9 fwd exact state
 * (throughput)), chunk length, waves per pair,
 * dynamic LDS bytes.  Pure function, needs no device.  (Reported for tensors whose rows and planes start on 128-byte
 * lines -- M a multiple of 32; other launches use the "general pitch" instantiations of the same builds, ids 11-20.
 * Ids 21-28: the throughput builds with the bridge between workgroups, see sdp_plan_parts.) */
int sdp_plan(int pass, int B, int N, int M, int has_lens, int exact_state, int cus, int *kernel_id, int *chunk,
             int *waves, size_t *lds);

/* ... and whether that launch would spread every pair over several workgroups (CUs): the number of 64-row strips per
 * workgroup, or 0 for one workgroup per pair.  It is done where it was measured to pay (round 5's table, profiles/
 * r05_parts_table.txt): the FORWARD sweep of padded batches with per-pair lengths that do not outnumber the CUs (the batch
 * takes as long as its longest pair) from pairs of more than eight strips on -- the backward sweep with per-pair lengths keeps
 * one workgroup per pair since round 5 --, and the backward sweep of a few EQUAL pairs (<= CUs / 4) of more than twelve strips;
 * never the adjoint pair.  The boundary between two parts of a pair then crosses
 * CUs through 8-byte granules in the tail of the state buffer.  Results do not depend on it (bit-identical).  Such
 * launches wait for the previous one of their kind on the same device, whatever its stream (two of them sharing the chip
 * could starve each other's producers); during stream capture that ordering is the graph's / the caller's. */
int sdp_plan_parts(int pass, int B, int N, int M, int has_lens, int exact_state, int cus);

#ifdef SDP_EXPERIMENTS
/* Only in libraries built with -DSDP_EXPERIMENTS (never the shipped one): timing experiments that produce WRONG
 * results.  bit0/1/2: inputs / outputs / state of every pair alias pair 0 (all traffic cache-served); bit3: strips
 * never publish their progress, so every hand-off times out (tests the SDP_E_HANDOFF path); 16 / 32: scores kernel choice;
 * 64: never spread a pair over several workgroups (128 / 256: not in the backward / forward sweep); 512: wherever possible;
 * 1024: sdp_set_trace stamps the backward sweep instead of the forward; 2048: backward of the scores with dS in a pass of its own;
 * 4096: the backward sweep runs the steps of all-zero chunks too (A/B of the exact-zero skip; same results).
 * Returns the old mask. */
int sdp_set_debug(int mask);
/* Cycle stamps of the forward sweep (tools/fwd_trace.py): a device buffer of >= 40 KiB, or NULL to switch it off. */
int sdp_set_trace(void *buf);
#endif

#ifdef __cplusplus
}
#endif
#endif /* SDP_H_ */
