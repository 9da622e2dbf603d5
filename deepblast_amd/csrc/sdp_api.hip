// sdp_api.hip -- host side of the C ABI declared in include/sdp.h.
// Validates arguments, sizes the launch and enqueues one kernel per call on the
// caller's stream.  No device allocation, no synchronisation.  Process-wide state: one 64-byte block of
// host-pinned status words per device (created on the first launch there), through which a kernel reports a
// strip hand-off that timed out; the experiment switches (sdp_set_debug) exist only in -DSDP_EXPERIMENTS builds.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "sdp_kernels.h"

namespace {

thread_local char g_err[256] = "";
#ifdef SDP_EXPERIMENTS
std::atomic<int> g_dbg{0};
std::atomic<unsigned long long *> g_trace{nullptr};
#endif

// ---- per-device status words (host-pinned, device-visible) ----
// Published once per device under g_status_mu and read through atomics afterwards: launches from several host threads
// may race with the first launch on a device.  Created by sdp_init(device) -- or by the first launch there; a caller
// that captures launches into a hipGraph calls sdp_init first (an allocation inside a capture fails, and the launch
// would then run without a way to report a hand-off time-out).
constexpr int MAX_DEV = 64;
std::mutex g_status_mu;
std::atomic<int *> g_status_host[MAX_DEV];
std::atomic<int *> g_status_dev[MAX_DEV];
std::atomic<int> g_status_seen[MAX_DEV];  // time-outs already reported to the caller

int *status_words(int device)  // device pointer, or nullptr (then kernels cannot report)
{
    if (device < 0 || device >= MAX_DEV) return nullptr;
    if (int *d = g_status_dev[device].load(std::memory_order_acquire)) return d;
    std::lock_guard<std::mutex> lk(g_status_mu);
    if (int *d = g_status_dev[device].load(std::memory_order_relaxed)) return d;
    int *h = nullptr, *d = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    memset(h, 0, 64);
    if (hipHostGetDevicePointer((void **)&d, h, 0) != hipSuccess) {
        (void)hipHostFree(h);
        return nullptr;
    }
    g_status_seen[device].store(0);
    g_status_host[device].store(h, std::memory_order_release);
    g_status_dev[device].store(d, std::memory_order_release);
    return d;
}

int fail(int code, const char *msg)
{
    snprintf(g_err, sizeof(g_err), "%s", msg);
    return code;
}

int fail_hip(hipError_t e, const char *what)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

struct Variant {
    const void *kernel;
    int K, maxw, id;
};

// kernel builds: [0] fwd (throughput: K=32, <= 4 waves), [1] bwd (throughput: K=32, <= 4 waves), [2] adj-fwd,
// [3] adj-bwd, [4] bwd (latency: K=16, <= 8 waves), [5] fwd writing the exact (float2) state for the adjoint
// sweeps, [6] fwd (latency: K=16, <= 8 waves), [7] / [8] bwd reading the exact state (throughput / latency), [9] fwd writing the exact state (throughput)
Variant variant(int id)
{
    switch (id) {
    case 0: return {(const void *)sdp_fwd_kernel, SDP_K_FWD, SDP_MAXW_FWD, 0};
    case 1: return {(const void *)sdp_bwd_kernel, SDP_K_BWD, SDP_MAXW_BWD_Q, 1};
    case 2: return {(const void *)sdp_adj_fwd_kernel, SDP_K_AFWD, SDP_MAXW_AFWD, 2};
    case 3: return {(const void *)sdp_adj_bwd_kernel, SDP_K_ABWD, SDP_MAXW_ABWD, 3};
    case 5: return {(const void *)sdp_fwd_x_kernel, SDP_K_FWD_LAT, SDP_MAXW_FWD_LAT, 5};
    case 6: return {(const void *)sdp_fwd_lat_kernel, SDP_K_FWD_LAT, SDP_MAXW_FWD_LAT, 6};
    case 7: return {(const void *)sdp_bwd_x_kernel, SDP_K_BWD, SDP_MAXW_BWD, 7};
    case 8: return {(const void *)sdp_bwd_x_lat_kernel, SDP_K_BWD_LAT, SDP_MAXW_BWD_LAT, 8};
    case 9: return {(const void *)sdp_fwd_x_tp_kernel, SDP_K_FWD, SDP_MAXW_FWD, 9};
    case 10: return {(const void *)sdp_adj_fwd_loss_kernel, SDP_K_AFWD, 4, 10};  // adj-fwd with the loss seed formed in the kernel
    // general-pitch instantiations (staged blocks aligned to lines of memory through run-time per-row offsets): id + 11
    case 11: return {(const void *)sdp_fwd_g_kernel, SDP_K_FWD, SDP_MAXW_FWD, 11};
    case 12: return {(const void *)sdp_bwd_g_kernel, SDP_K_BWD, SDP_MAXW_BWD_Q, 12};   // (general pitch: 210 registers since round 6 -- the flush's index arrays are gone -- so eight waves fit, like the aligned build's)
    case 14: return {(const void *)sdp_adj_bwd_g_kernel, SDP_K_ABWD, SDP_MAXW_ABWD, 14};
    case 15: return {(const void *)sdp_bwd_lat_g_kernel, SDP_K_BWD_LAT, SDP_MAXW_BWD_LAT, 15};
    case 18: return {(const void *)sdp_bwd_x_g_kernel, SDP_K_BWD, SDP_MAXW_BWD, 18};
    case 19: return {(const void *)sdp_bwd_x_lat_g_kernel, SDP_K_BWD_LAT, SDP_MAXW_BWD_LAT, 19};
    case 20: return {(const void *)sdp_fwd_x_tp_g_kernel, SDP_K_FWD, SDP_MAXW_FWD, 20};
    // PARTS instantiations of the throughput builds (a pair over several workgroups): 21 + (fwd, fwd exact, bwd, bwd exact), + 4 general pitch
    case 21: return {(const void *)sdp_fwd_p_kernel, SDP_K_FWD, SDP_MAXW_FWD, 21};
    case 22: return {(const void *)sdp_fwd_x_tp_p_kernel, SDP_K_FWD, SDP_MAXW_FWD, 22};
    case 23: return {(const void *)sdp_bwd_p_kernel, SDP_K_BWD, SDP_MAXW_BWD, 23};
    case 24: return {(const void *)sdp_bwd_x_p_kernel, SDP_K_BWD, SDP_MAXW_BWD, 24};
    case 25: return {(const void *)sdp_fwd_pg_kernel, SDP_K_FWD, SDP_MAXW_FWD, 25};
    case 26: return {(const void *)sdp_fwd_x_tp_pg_kernel, SDP_K_FWD, SDP_MAXW_FWD, 26};
    case 27: return {(const void *)sdp_bwd_pg_kernel, SDP_K_BWD, SDP_MAXW_BWD, 27};
    case 28: return {(const void *)sdp_bwd_x_pg_kernel, SDP_K_BWD, SDP_MAXW_BWD, 28};
    case 36: return {(const void *)sdp_bwd_pipe_kernel, SDP_K_BWD, SDP_MAXW_BWD_Q, 36};   // [1] with the chunk as one software pipeline (long pairs)
    // [0] / [9] with the cleaning of what lies beside the matrix (per-pair lengths, partial strips; sdp_kernels.hip "need_clean")
    case 37: return {(const void *)sdp_fwd_c_kernel, SDP_K_FWD, SDP_MAXW_FWD, 37};
    case 38: return {(const void *)sdp_fwd_x_tp_c_kernel, SDP_K_FWD, SDP_MAXW_FWD, 38};
    case 39: return {(const void *)sdp_fwd_lat_c_kernel, SDP_K_FWD_LAT, SDP_MAXW_FWD_LAT, 39};   // [6] ...
    case 40: return {(const void *)sdp_fwd_x_c_kernel, SDP_K_FWD_LAT, SDP_MAXW_FWD_LAT, 40};     // [5] ...
    default: return {(const void *)sdp_bwd_lat_kernel, SDP_K_BWD_LAT, SDP_MAXW_BWD_LAT, 4};
    }
}

int num_cus(int device)
{
    static thread_local int cached[64] = {0};
    if (device >= 0 && device < 64 && cached[device] > 0) return cached[device];
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || n <= 0) n = 256;
    if (device >= 0 && device < 64) cached[device] = n;
    return n;
}

size_t lds_bytes(int pass, int K, int W, int mcap, size_t *stage_off, int nin = 0)
{
    const int nslot = 2;  // boundary rows are shared by alternate strips (see the kernel)
    size_t off = (size_t)nslot * mcap * sdp::boundary_slot_bytes(pass) + 64 + (size_t)nslot * sdp::FRAME_CAP * 4;  // boundary rows, progress words, frame words
    off = ((off + 15) & ~(size_t)15) + 16;   // (+ 16: the staged-output ring of wave 0 writes one float in front of itself, see FLUSH2)
    if (stage_off) *stage_off = off;
    return off + (size_t)W * sdp::stage_floats(pass, K, nin) * sizeof(float);
}

int check_shape(int B, int N, int M, int variant)
{
    if (B <= 0 || N <= 0 || M <= 0) return fail(SDP_E_SHAPE, "B, N and M must be positive");
    if (M > sdp::MAX_COLS) return fail(SDP_E_MAXCOLS, "M exceeds sdp_max_cols()");
    if (variant != SDP_NW && variant != SDP_SW) return fail(SDP_E_VARIANT, "variant must be SDP_NW or SDP_SW");
    if ((size_t)N * (size_t)M > ((size_t)1 << 28)) return fail(SDP_E_TOOBIG, "N*M exceeds 2^28 cells per pair");
    return 0;
}

// What a launch will use: which kernel build, how many waves per pair, how much LDS.  Pure function of the
// problem and the CU count (no device access), so that the policy can be tested without a GPU (sdp_plan).
struct Plan {
    Variant v;
    int W;
    size_t lds, stage_off;
    int parts;   // strips per workgroup when the pairs are spread over several workgroups (0: one workgroup per pair)
};

// kernel id -> its general-pitch instantiation (or itself if it has none: builds whose staged blocks keep the
// column-aligned geometry -- latency forward builds, the adjoint forward)
int general_id(int id)
{
    switch (id) {
    case 0: case 37: return 11;
    case 1: return 12;
    case 3: return 14;
    case 4: return 15;
    case 7: return 18;
    case 8: return 19;
    case 9: case 38: return 20;
    default: return id;
    }
}

// Strips per part when a pair is spread over several workgroups (sdp_kernels.hip, "PARTS"): one strip per wave of the
// 4-wave throughput builds.
constexpr int PART_STRIPS = 4;
inline int parts_per_pair(int N) { return (sdp::state_nstrips(N) + PART_STRIPS - 1) / PART_STRIPS; }

Plan plan(int pass, int B, int N, int M, bool has_lens, bool exact_state, int cus, int forced_waves, bool fused_seed = false,
          bool general_pitch = false, int allow_parts = 1 /* 0 never, 1 where it pays, 2 wherever it is possible */)
{
    const int nstrips = sdp::state_nstrips(N);
    const int mcap = (M + 63) / 64 * 64;
    // Waves per pair.  A batch that occupies the GPU is bound by HBM/fabric traffic and runs best with one wave
    // per SIMD and long chunks; a smaller batch is bound by the length of the strip pipeline of a single pair,
    // which more waves (and the shorter-chunk latency builds) shorten.
    // With per-pair lengths the longest pairs set the time of a batch that does not queue up on the CUs, so such
    // a batch is treated like a small one (measured, 256 pairs with n, m ~ U[64,1024]: 1.27 ms vs 1.41 ms).
    const bool sweep12 = pass == sdp::PASS_FWD || pass == sdp::PASS_BWD;
    // measured crossover (round 5, 512 x 512, 256 CUs; profiles/r05_small_batches.txt): the forward sweep's latency build holds 133 us
    // up to 64 pairs and is at 170-190 us from 80 on, where the throughput build takes 148-151 -- ~72 pairs; the backward sweep's
    // 8-wave form is level with its 4-wave form up to 128 pairs
    const bool full = (pass == sdp::PASS_FWD ? B * 7 >= cus * 2 : B * 2 >= cus) && !(has_lens && B <= 2 * cus);
    Variant v = variant(pass);
    int W = forced_waves;
    if (W <= 0) {
        W = (sweep12 || pass == sdp::PASS_AFWD) ? (full ? 4 : 8) : SDP_DEFAULT_WAVES;  // (adj-bwd is compiled for <= 4)
        // More pairs than CUs: with 2 waves two workgroups share a CU (their LDS fits twice) and overlap each
        // other's ramps -- a round of 2*CUs pairs then takes 1.8x a 4-wave round of CUs pairs.  Pick the cheaper.
        const int r4 = (B + cus - 1) / cus, r2 = (B + 2 * cus - 1) / (2 * cus);
        if (full && sweep12 && B > cus && 181 * r2 < 100 * r4) W = 2;
    }
    if (sweep12) {
        // the throughput builds need 4 waves' worth of LDS for their longer chunks; fall back to the latency
        // builds when that does not fit (long M) or when more waves are wanted
        const int w4 = nstrips < 4 ? nstrips : 4;
        // (the packed backward build runs up to 8 waves since round 5, its general-pitch twin since round 6 -- the exact-state twins
        //  only 4: a launch that will end up in one of those is judged by their limit)
        const int maxw = (pass == sdp::PASS_BWD && exact_state) ? SDP_MAXW_BWD : v.maxw;
        if (W > maxw || lds_bytes(pass, v.K, w4, mcap, nullptr) > 160 * 1024) v = variant(pass == sdp::PASS_FWD ? 6 : 4);
    }
    if (pass == sdp::PASS_FWD && exact_state) v = variant(v.id == 0 ? 9 : 5);
    if (pass == sdp::PASS_BWD && exact_state) v = variant(v.id == 1 ? 7 : 8);
    const int nin = (pass == sdp::PASS_AFWD && fused_seed) ? 3 : 0;   // three staged planes (ref, pred, G)
    if (nin) v = variant(10);
    // the aligned-pitch forward builds come without any edge cleaning (sdp_kernels.hip "need_clean"); per-pair lengths or partial
    // strips take their twins that carry it
    if (pass == sdp::PASS_FWD && (has_lens || (N & 63) != 0)) {
        if (v.id == 0) v = variant(37);
        else if (v.id == 9) v = variant(38);
        else if (v.id == 6) v = variant(39);
        else if (v.id == 5) v = variant(40);
    }
    if (general_pitch) v = variant(general_id(v.id));
    // The packed backward sweep's pipelined twin (sdp_kernels.hip, sdp_bwd_pipe_kernel) trades instruction issue for memory
    // latency.  Steady-state A/B over 24 shapes (tools/steady.py, +- 0.3 us; profiles/r05_steady_pipe.txt): it pays 2.3 % where
    // every CU holds ONE pair of long rows (256 x 1024^2, 256 x 512 x 1024), is level at 256 x 768 x 640 / 300 x 2000 / 192 x 1024^2, and
    // costs 1.1-3.7 % everywhere else -- shorter rows (the headline 256 x 512^2: 279.0 vs 272.5 us; 256 x 2048 x 256: 2.7 %), fewer
    // pairs than CUs (128 x 1024^2), more (384 x 1024^2, 512 x 768^2), 8 waves per pair (64 x 512^2).  So: only there.
    const bool bwd_pipe_pays = W == 4 && B <= cus && B * 8 >= cus * 7 && M >= 1024 && (long long)N * M >= 450000;
    if (v.id == 1 && bwd_pipe_pays) v = variant(36);
    // A pair over several workgroups (sdp_kernels.hip, "PARTS"): parts of four strips, each on a CU of its own, one strip
    // per wave of the 4-wave throughput builds, instead of one CU taking all the pair's strips in rounds.  It pays only
    // where CUs would otherwise idle AND the pair is long enough for the extra lag per bridge (measured, round 3, us
    // forward / backward, one workgroup per pair -> parts):
    //   per-pair lengths, 256 pairs: n, m <= 1022 x 1020 (BASELINE configs[2]) 640 / 610 -> 515 / 465; <= 640^2 301 / 245 ->
    //   287 / 240; <= 512^2 (two parts) 199 / 169 -> 210 / 148; 128 pairs <= 512^2 170 / 157 -> 194 / 146; 64 pairs <= 1022 x 1020
    //   532 / 385 -> 462 / 309; more pairs than CUs, <= 1022 x 1020: 384 pairs 752 / 761 -> 778 / 743, 512 pairs 832 / 1023 ->
    //   1000 / 950, 700 pairs 998 / 1188 -> 1250 / 1300;
    //   equal pairs: 16 x 1024^2 445 / 330 -> 451 / 291; 64 x 640 x 500 240 / 163 -> 260 / 183 (worse); 16 x 512^2 worse.
    // So (round 3; round 5's re-measurement below drops the backward sweep with per-pair lengths): with per-pair lengths whenever
    // the batch does not outnumber the CUs -- the forward sweep from three parts on, the backward sweep from two; equal pairs
    // only the backward sweep of pairs of four parts.  The adjoint pair (float64
    // carries) keeps one workgroup per pair.
    int parts = 0;
    const bool parts_fit = sweep12 && forced_waves <= 0 && nstrips > PART_STRIPS;
    // Round 5, re-measured after the backward sweep's changes (profiles/r05_parts_table.txt; us, one workgroup per pair -> parts):
    // with per-pair lengths the BACKWARD sweep no longer gains from parts anywhere but at BASELINE configs[2] (540 -> 512-522,
    // and 515 when only the forward sweep uses them) and loses elsewhere (<= 640^2: 213 -> 236, <= 512^2: 146 -> 160, 64 pairs
    // <= 1022 x 1020: 336 -> 355, 32 pairs <= 2000 x 1000: 528 -> 582): dropped.  The forward sweep keeps them from three parts
    // on (<= 1022 x 1020: 600 -> 514, <= 640^2: 277 -> 257, 64 pairs: 510 -> 400; <= 512^2: 176 -> 195, not taken), equal pairs
    // the backward sweep of a few pairs of more than twelve strips (16 x 1024^2: 303 -> 279).
    const bool parts_pay = B <= cus && (has_lens ? (pass == sdp::PASS_FWD && nstrips > 2 * PART_STRIPS)
                                                 : (pass == sdp::PASS_BWD && nstrips > 3 * PART_STRIPS && (long long)B * parts_per_pair(N) <= cus));
    if (parts_fit && (allow_parts == 2 || (allow_parts == 1 && parts_pay))) {
        const Variant tv = variant(21 + (pass == sdp::PASS_FWD ? 0 : 2) + (exact_state ? 1 : 0) + (general_pitch ? 4 : 0));
        if (lds_bytes(pass, tv.K, PART_STRIPS, mcap, nullptr) <= 160 * 1024) {
            v = tv;
            W = PART_STRIPS;
            parts = PART_STRIPS;
        }
    }
    if (W > v.maxw) W = v.maxw;
    if (W > nstrips) W = nstrips;
    size_t off = 0, lds = 0;
    for (;; --W) {  // fewer waves if the boundary rows (long M) plus staging exceed the 160 KiB of LDS
        lds = lds_bytes(pass, v.K, W, mcap, &off, nin);
        if (lds <= 160 * 1024 || W == 1) break;
    }
    // Two-wave workgroups (more pairs than CUs) are meant to share a CU in PAIRS, one wave per SIMD.  Since round 5 the backward
    // build needs fewer than 256 registers, so the hardware would also take four of them -- two waves per SIMD, each at half
    // speed, other CUs short of work: 247 -> 262 us at 512 x 512^2.  Asking for half a CU's LDS keeps it at two.
    if (sweep12 && W == 2 && B > cus && lds < 80 * 1024) lds = 80 * 1024;
    return {v, W, lds, off, parts};
}

// Where the 32-step units of the skewed state live (sdp_kernels.hip, "Skewed state addressing"): every (pair, strip) is one
// contiguous stream of units.  (A "marching" layout -- unit u of every (pair, strip) in one slab, so that a batch whose pairs
// advance in lockstep sweeps memory front to back -- measured equal in round 3 and was removed in round 6: the sweeps are not
// bound by the order in which memory is visited.  The two strides stay parameters of the kernels.)
// All four sweeps of a problem (B, N, M) derive the same layout from the same three numbers.
void state_layout(sdp::Params &p)
{
    const size_t units = (size_t)p.tpad / sdp::STATE_UNIT_STEPS;
    p.st_ps = units * sdp::STATEQ_UNIT_BYTES;
    p.st_us = sdp::STATEQ_UNIT_BYTES;
    p.st2_ps = units * sdp::STATE2_UNIT_BYTES;
    p.st2_us = sdp::STATE2_UNIT_BYTES;
}

// A kernel of an EARLIER call on this device gave up waiting for a strip hand-off: its results are wrong.  Reported
// once, by the next call on the device (or by sdp_device_status), as SDP_E_HANDOFF.
int pending_handoff_error(int device)
{
    if (device < 0 || device >= MAX_DEV) return 0;
    const volatile int *h = g_status_host[device].load(std::memory_order_acquire);
    if (!h) return 0;
    const int n = h[0];
    // exactly one caller reports a given count: the one whose exchange moves `seen` to it
    int seen = g_status_seen[device].load();
    do {
        if (n == seen) return 0;
    } while (!g_status_seen[device].compare_exchange_weak(seen, n));
    snprintf(g_err, sizeof(g_err),
             "a strip hand-off timed out in an earlier launch on device %d (%d so far; first: pair %d, strip %d, chunk %d, "
             "pass %d): the results of that launch are invalid",
             device, n, h[1], h[2], h[3] & 0xffffff, (h[3] >> 24) & 0xff);
    return SDP_E_HANDOFF;
}

// Variable-length batches with more pairs than CUs run longest-first.  The order is computed by the forward call
// (sdp_order_kernel) into the tail of the state buffer it fills, and the other sweeps -- which receive that state
// and must be given the same lengths -- read it from there.
size_t packed_body_bytes(int B, int N, int M)
{
    return (size_t)B * sdp::state_nstrips(N) * (sdp::state_tpad(M) / sdp::STATE_UNIT_STEPS) * sdp::STATEQ_UNIT_BYTES;
}
bool wants_order(int B, int N, const int32_t *lens, int device)
{
    (void)N;
    return lens != nullptr && B > num_cus(device);
}
const int *order_in_state(const void *state, int B, int N, int M, bool exact)
{
    const size_t body = exact ? (size_t)B * sdp::state_rows2(N, M) * 64 * sizeof(float2)
                              : packed_body_bytes(B, N, M);
    return reinterpret_cast<const int *>(static_cast<const char *>(state) + body);
}
// ... and behind the order: the bridge rows between the parts of a pair (8-byte granules; reset before every launch
// that uses them, so the forward and the backward sweep share the room)
size_t bridge_bytes(int B, int N, int M)
{
    const int np = parts_per_pair(N);
    return np > 1 ? (size_t)B * (np - 1) * sdp::xb_row_granules(M) * 8 : 0;
}
// ... and behind those the dispatch order of the parts of a batch with per-pair lengths (one int per workgroup)
size_t parts_map_bytes(int B, int N)
{
    const int np = parts_per_pair(N);
    return np > 1 ? ((size_t)B * np * 4 + 255) / 256 * 256 : 0;
}
unsigned long long *bridge_in_state(const void *state, int B, int N, int M, bool exact)
{
    return reinterpret_cast<unsigned long long *>(const_cast<char *>(reinterpret_cast<const char *>(order_in_state(state, B, N, M, exact))) +
                                                  sdp::state_order_bytes(B));
}

// flags or-ed into `variant` (include/sdp.h): SDP_EXACT_STATE, SDP_WAVES(w)
struct VariantBits {
    int variant, waves;
    bool exact, et_bcast, ref;
    int flags;   // Params::flags: bit 0 SDP_NO_ZERO_SKIP, bit 1 SDP_NO_FILL
};
// The packed state keeps two 20-bit weights per cell (absolute error <= 2^-21 per weight, sdp_kernels.hip "The packed state").  Its
// rounding error travels along an alignment path like a random walk (the 24-bit format of rounds 1-3 instead lost 1.7e-8 of
// E per step of a SATURATED path, 7.5e-5 at N = M = 2048: the 20-bit fields decode a saturated weight to exactly 1 and do
// not have that term); tests/test_parity_gpu.py::test_packed_state_at_the_longest_paths_it_serves holds max |dE| at
// N = M = 2048 (N + M = 4096, the longest problem the packed state serves) on soft, steep and peaked scores to half the
// 1e-4 bound.  Longer problems always use the exact (float2) state, whose largest weight is the complement of the other
// two (q_sharpen): <= 1e-5 at N = 60000.  sdp_state_bytes covers either layout; forward and backward apply the same rule.
constexpr int PACKED_MAX_PATH = 4096;
// Thin problems as well (round 5): with fewer than 32 rows (or columns) the rounding errors of the packed weights along the long
// axis do not average out over many paths -- flat scores, max |dE| by shape, packed (exact), tools/thin_probe.py / profiles/
// r05_thin.txt: 2 x 2048 1.0e-4 (4.5e-6), 4 x 2048 7.8e-5, 8 x 2048 6.1e-5, 16 x 1772 5.0e-5, 32 x 2048 3.6e-5, 64 x 2048 1.6e-5;
// any N x 512 <= 3.0e-5.  The soak of round 5 met 1.17e-4 at 2 x 1772 (positive gap scores).  Such problems are tiny: the exact state.
// Round 6: the same holds for a thin PAIR inside a fat padded batch with per-pair lengths (the reference's inference loop slices
// every pair and calls the decoder on the slice, alignment.py:165-170: each pair gets what a call of its own shape would get).
// The state format is a property of the kernel build, so such pairs are ROUTED: a launch with per-pair lengths whose padded shape
// admits thin long pairs (sdp::thin_pair: min(n, m) < 32 and max(n, m) > 512) runs the packed-state build with those pairs
// skipped (Params::route = 1) and, behind it, the exact-state build for those pairs only (route = 2; its workgroups of all other
// pairs leave at once).  The thin pair's float2 state lives inside the pair's own packed record: strip s starts where the packed
// strip s starts (st2_ps = st_ps) -- a wide thin pair has one strip and needs <= tpad x 512 B of the nstrips x tpad x 320 B, a tall
// thin one <= 3 units x 16 KB per strip of the tpad x 320 B a packed strip owns -- which fits as soon as the padded shape has more
// than 64 rows and more than 65 columns; padded shapes below that (and long) take the exact state as a whole, like thin problems.
inline bool exact_for(bool flag, int N, int M)
{
    const int lo = N < M ? N : M, hi = N < M ? M : N;
    return flag || N + M > PACKED_MAX_PATH || (lo < sdp::THIN_FITS && hi > sdp::THIN_HI);
}
// does a launch of this padded shape with per-pair lengths have to route thin pairs to the exact-state build?
inline bool routes_thin(bool exact, int N, int M, const int32_t *lens)
{
#ifdef SDP_EXPERIMENTS
    if (g_dbg.load() & 32768) return false;   // sdp_set_debug(32768): no routing (A/B timing of the second launch; thin pairs then keep the packed state)
#endif
    return lens != nullptr && !exact && (N > sdp::THIN_HI || M > sdp::THIN_HI);
}

VariantBits split_variant(int variant)
{
    VariantBits v;
    v.exact = (variant & SDP_EXACT_STATE) != 0;
    v.et_bcast = (variant & SDP_ET_BROADCAST) != 0;
    v.waves = (variant >> 12) & 0xf;
    v.ref = (variant & SDP_REF_ROUNDING) != 0;
    v.flags = ((variant & SDP_NO_ZERO_SKIP) ? 1 : 0) | ((variant & SDP_NO_FILL) ? 2 : 0);
    v.variant = variant & ~(SDP_EXACT_STATE | SDP_ET_BROADCAST | SDP_REF_ROUNDING | SDP_NO_ZERO_SKIP | SDP_NO_FILL | (0xf << 12));
    return v;
}

// raise the dynamic-LDS limit once per (thread, device, kernel) -- it is sticky, and the value is the
// 160 KiB the hardware has, so concurrent callers cannot disagree
int raise_lds_limit(const Variant &v, int device)
{
    static thread_local unsigned long long lds_raised[41] = {0};  // per kernel id: bit d = done on device d
    if (device >= 64 || !(lds_raised[v.id] >> device & 1ull)) {
        hipError_t e = hipFuncSetAttribute(v.kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return fail_hip(e, "hipFuncSetAttribute(MaxDynamicSharedMemorySize)");
        if (device < 64) lds_raised[v.id] |= 1ull << device;
    }
    return 0;
}

int launch(int pass, sdp::Params &p, int device, void *stream, bool exact_state = false, int forced_waves = 0, bool fused_seed = false,
           const void *state = nullptr, int route = 0)
{
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
    if (int rc = pending_handoff_error(device)) return rc;
    p.nstrips_max = sdp::state_nstrips(p.N);
    p.tpad = sdp::state_tpad(p.M);
    p.mcap = (p.M + 63) / 64 * 64;
    state_layout(p);
    p.route = route;
    if (route == 2) {   // thin pairs only, exact-state build, inside the packed records (see exact_for): one workgroup per pair, batch order
        p.st2_ps = p.st_ps;
        p.order = nullptr;
        state = nullptr;
    }
    p.status = status_words(device);
#ifdef SDP_EXPERIMENTS
    p.dbg = g_dbg.load();
    p.trace = g_trace.load();
#else
    p.dbg = 0;
#endif
    // rows and planes of the staged tensors on 128-byte lines?  (M a multiple of 32 makes every plane offset b*N*M a
    // multiple of 32 floats too; the tensors themselves come from an allocator that aligns far beyond 128 bytes, but a
    // caller may pass a view)
    auto misaligned = [](const void *ptr) { return ptr != nullptr && ((uintptr_t)ptr & 127u) != 0; };
    const bool general_pitch = (p.M & 31) != 0 || misaligned(p.sin0) || misaligned(p.sin1) || misaligned(p.sin2) || misaligned(p.sout);
    int allow_parts = state != nullptr ? 1 : 0;
#ifdef SDP_EXPERIMENTS
    if (allow_parts && (g_dbg.load() & 512)) allow_parts = 2;   // sdp_set_debug(512): parts wherever they are possible (bit-identity test)
    if (g_dbg.load() & 64) allow_parts = 0;   // sdp_set_debug(64): one workgroup per pair whatever the shape (A/B timing, bit-identity test)
    if ((g_dbg.load() & 128) && pass == sdp::PASS_BWD) allow_parts = 0;   // (128: ... in the backward sweep only)
    if ((g_dbg.load() & 256) && pass == sdp::PASS_FWD) allow_parts = 0;   // (256: ... in the forward sweep only)
#endif
    const Plan pl = plan(pass, p.B, p.N, p.M, p.lens != nullptr, exact_state, num_cus(device), forced_waves, fused_seed, general_pitch, allow_parts);
    const Variant v = pl.v;
    const int W = pl.W;
    const size_t lds = pl.lds, off = pl.stage_off;
    p.stage_off = (int)off;
    if (int rc = raise_lds_limit(v, device)) return rc;
    unsigned grid = (unsigned)p.B;
    if (pl.parts) {
        p.parts = pl.parts;
        p.nparts_max = parts_per_pair(p.N);
        p.xb = bridge_in_state(state, p.B, p.N, p.M, exact_state);
        p.xb_row = sdp::xb_row_granules(p.M);
        // every granule "not written yet" (tag 0x7f7f7f7f): enqueued on the caller's stream like the launch itself.  A
        // kernel of our own, not hipMemsetAsync: captured into a graph (torch.cuda.graph) the memset node did not take effect
        // before the sweep on replay -- consumers then took whatever the buffer held for granules (NaN results; tests/
        // test_robustness_gpu.py::test_sweeps_can_be_captured_in_a_graph).
        {
            const size_t n8 = bridge_bytes(p.B, p.N, p.M) / 8;
            const unsigned blocks = (unsigned)((n8 + 1023) / 1024 < 4096 ? (n8 + 1023) / 1024 : 4096);
            hipLaunchKernelGGL(sdp_bridge_reset_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p.xb, n8);
            e = hipGetLastError();
            if (e != hipSuccess) return fail_hip(e, "sdp_bridge_reset_kernel");
        }
        grid = (unsigned)p.B * (unsigned)p.nparts_max;
        p.order = nullptr;
        if (p.lens != nullptr) {
            int *map = reinterpret_cast<int *>(reinterpret_cast<char *>(p.xb) + bridge_bytes(p.B, p.N, p.M));
            hipLaunchKernelGGL(sdp_parts_map_kernel, dim3((grid + 3) / 4), dim3(256), 0, (hipStream_t)stream, p.lens, map, p.B, p.N, p.M,
                               p.nparts_max, pl.parts);
            e = hipGetLastError();
            if (e != hipSuccess) return fail_hip(e, "sdp_parts_map_kernel");
            p.wg_map = map;
        }
    }
    void *args[] = {&p};
    if (pl.parts) {
        // Two parts launches must not share the chip.  Within ONE launch a consumer can never keep its producer off a CU
        // (index order, see sdp_parts_map_kernel); with two launches on two streams the waiting parts of one could hold
        // the CUs the other's producers are queued for, and vice versa -- a deadlock that the bounded spins would turn
        // into SDP_E_HANDOFF and wrong results.  So every parts launch waits for the previous one on this device,
        // whatever stream that was on (an event per device; the lock keeps wait + launch + record of two host threads
        // apart).  Not during stream capture, where an event of uncaptured work may not be waited on: launches inside one
        // graph are ordered by the graph, and concurrent graphs that both hold parts launches are the caller's to order.
        static std::mutex mtx;
        static hipEvent_t last[64];
        static bool have[64] = {false};
        std::lock_guard<std::mutex> lock(mtx);
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        const bool track = device < 64 && hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess && cap == hipStreamCaptureStatusNone;
        if (track && have[device]) {
            e = hipStreamWaitEvent((hipStream_t)stream, last[device], 0);
            if (e != hipSuccess) return fail_hip(e, "hipStreamWaitEvent(previous parts launch)");
        }
        e = hipLaunchKernel(v.kernel, dim3(grid), dim3(64 * W), args, lds, (hipStream_t)stream);
        if (e != hipSuccess) return fail_hip(e, "hipLaunchKernel");
        if (track) {
            if (!have[device]) {
                e = hipEventCreateWithFlags(&last[device], hipEventDisableTiming);
                if (e != hipSuccess) return fail_hip(e, "hipEventCreateWithFlags");
                have[device] = true;
            }
            e = hipEventRecord(last[device], (hipStream_t)stream);
            if (e != hipSuccess) return fail_hip(e, "hipEventRecord(parts launch)");
        }
        return 0;
    }
    e = hipLaunchKernel(v.kernel, dim3(grid), dim3(64 * W), args, lds, (hipStream_t)stream);
    if (e != hipSuccess) return fail_hip(e, "hipLaunchKernel");
    return 0;
}

// dynamic-LDS limits of the scores kernels, once per (thread, device): sticky attributes, set before any launch that
// may be captured (sdp_init) or lazily by the first call
int raise_scores_limits(int device)
{
    static thread_local unsigned long long raised = 0;
    if (device < 64 && (raised >> device & 1ull)) return 0;
    const struct { const void *f; int bytes; const char *what; } ks[] = {
        {(const void *)sdp_scores_kernel, sdp::SCORES_LDS_BYTES, "hipFuncSetAttribute(sdp_scores_kernel)"},
        {(const void *)sdp_scores_x6_kernel, sdp::SCORES_X6_LDS_BYTES, "hipFuncSetAttribute(sdp_scores_x6_kernel)"},
        {(const void *)sdp_scores_x6s_kernel, sdp::SCORES_X6_LDS_BYTES, "hipFuncSetAttribute(sdp_scores_x6s_kernel)"},
        {(const void *)sdp_scores_x6w_kernel, sdp::SCORES_X6W_LDS_BYTES, "hipFuncSetAttribute(sdp_scores_x6w_kernel)"},
        {(const void *)sdp_scores_bwd_x_kernel, sdp::SCORES_X6W_LDS_BYTES, "hipFuncSetAttribute(sdp_scores_bwd_x_kernel)"},
        {(const void *)sdp_scores_bwd_y_kernel, sdp::SCORES_X6W_LDS_BYTES, "hipFuncSetAttribute(sdp_scores_bwd_y_kernel)"},
        {(const void *)sdp_scores_bwd_yf_kernel, sdp::SCORES_X6W_LDS_BYTES, "hipFuncSetAttribute(sdp_scores_bwd_yf_kernel)"},
    };
    for (const auto &k : ks) {
        const hipError_t e = hipFuncSetAttribute(k.f, hipFuncAttributeMaxDynamicSharedMemorySize, k.bytes);
        if (e != hipSuccess) return fail_hip(e, k.what);
    }
    if (device < 64) raised |= 1ull << device;
    return 0;
}

// variant | SDP_REF_ROUNDING (sdp_ref.hip): one workgroup of 256 threads per pair, three rolling anti-diagonals of float64
size_t ref_lds(int M) { return (size_t)3 * (M + 2) * sizeof(double); }
size_t ref_state_bytes(int B, int N, int M) { return (size_t)B * N * M * 3 * sizeof(float); }
int ref_prepare(int device)
{
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
    return pending_handoff_error(device);
}
int ref_launched(const char *what)
{
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : fail_hip(e, what);
}

}  // namespace

extern "C" {

int sdp_version(void) { return SDP_VERSION; }

const char *sdp_last_error_string(void) { return g_err; }

int sdp_max_cols(void) { return sdp::MAX_COLS; }

size_t sdp_state_bytes(int B, int N, int M)
{
    if (B <= 0 || N <= 0 || M <= 0) return 0;
    // 2 x 20 bits per cell, five dwords per four cells (ten rows of 1024 B per 32 steps of a strip); + the launch order of a
    // variable-length batch, the bridge rows and the dispatch map of a parts launch.  Problems longer than PACKED_MAX_PATH, and
    // thin long ones, keep the exact state (see exact_for).
    if (exact_for(false, N, M)) return sdp_state_d_bytes(B, N, M);
    return packed_body_bytes(B, N, M) + sdp::state_order_bytes(B) + bridge_bytes(B, N, M) + parts_map_bytes(B, N);
}

size_t sdp_state_d_bytes(int B, int N, int M)
{
    if (B <= 0 || N <= 0 || M <= 0) return 0;
    return (size_t)B * sdp::state_rows2(N, M) * 64 * sizeof(float2) + sdp::state_order_bytes(B) + bridge_bytes(B, N, M) + parts_map_bytes(B, N);
}

size_t sdp_state_bytes_v(int B, int N, int M, int variant)
{
    if (B <= 0 || N <= 0 || M <= 0) return 0;
    if (variant & SDP_REF_ROUNDING) return ref_state_bytes(B, N, M);
    return (variant & SDP_EXACT_STATE) ? sdp_state_d_bytes(B, N, M) : sdp_state_bytes(B, N, M);
}

size_t sdp_state_d_bytes_v(int B, int N, int M, int variant)
{
    if (B <= 0 || N <= 0 || M <= 0) return 0;
    return (variant & SDP_REF_ROUNDING) ? ref_state_bytes(B, N, M) : sdp_state_d_bytes(B, N, M);
}

int sdp_plan(int pass, int B, int N, int M, int has_lens, int exact_state, int cus, int *kernel_id, int *chunk,
             int *waves, size_t *lds)
{
    if (pass < 0 || pass > 3) return fail(SDP_E_VARIANT, "sdp_plan: pass must be 0..3");
    if (int rc = check_shape(B, N, M, SDP_NW)) return rc;
    if (cus <= 0) return fail(SDP_E_SHAPE, "sdp_plan: cus must be positive");
    const bool exact = (pass == sdp::PASS_FWD || pass == sdp::PASS_BWD) ? exact_for(exact_state != 0, N, M) : exact_state != 0;
    const Plan pl = plan(pass, B, N, M, has_lens != 0, exact, cus, 0);
    if (kernel_id) *kernel_id = pl.v.id;
    if (chunk) *chunk = pl.v.K;
    if (waves) *waves = pl.W;
    if (lds) *lds = pl.lds;
    return 0;
}

int sdp_plan_parts(int pass, int B, int N, int M, int has_lens, int exact_state, int cus)
{
    if (pass < 0 || pass > 3 || check_shape(B, N, M, SDP_NW) || cus <= 0) return 0;
    const bool exact = (pass == sdp::PASS_FWD || pass == sdp::PASS_BWD) ? exact_for(exact_state != 0, N, M) : exact_state != 0;
    return plan(pass, B, N, M, has_lens != 0, exact, cus, 0).parts;
}

int sdp_init(int device)
{
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
    if (device >= 0 && device < MAX_DEV && !status_words(device)) return fail(SDP_E_SELFTEST, "sdp_init: could not create the host-pinned status words");
    for (int id = 0; id <= 40; ++id) {   // (21-28: the parts instantiations, 36: the pipelined backward twin, 37-40: the cleaning forward twins)
        const Variant v = variant(id);
        if (v.id != id) continue;   // ids without a build of their own map to the default
        if (int rc = raise_lds_limit(v, device)) return rc;
    }
    return raise_scores_limits(device);
}

int sdp_device_status(int device, int32_t info[4])
{
    const volatile int *h = (device >= 0 && device < MAX_DEV) ? g_status_host[device].load(std::memory_order_acquire) : nullptr;
    if (!h) {
        if (info) info[0] = info[1] = info[2] = info[3] = 0;
        return 0;
    }
    if (info) info[0] = h[0], info[1] = h[1], info[2] = h[2], info[3] = h[3];
    return pending_handoff_error(device);
}

#ifdef SDP_EXPERIMENTS
int sdp_set_debug(int mask)
{
    return g_dbg.exchange(mask);
}
// device buffer (>= 4 pairs x 4 waves x 4 strip rounds or parts x 40 blocks x 8 stamps x 8 B = 160 KiB) that the forward sweep of pairs
// 0, 64, 128, 192 fills with shader-cycle stamps per 16-step block; null = off
int sdp_set_trace(void *buf)
{
    g_trace.store(static_cast<unsigned long long *>(buf));
    return 0;
}
#endif

int sdp_forward_f32(const float *theta, const float *A, float *state, float *Vt, int B, int N, int M,
                    const int32_t *lens, int variant, int device, void *stream)
{
    if (!theta || !A || !state || !Vt) return fail(SDP_E_NULLPTR, "sdp_forward_f32: null pointer");
    const VariantBits vb = split_variant(variant);
    const bool exact = exact_for(vb.exact, N, M);
    variant = vb.variant;
    if (int rc = check_shape(B, N, M, variant)) return rc;
    if (vb.ref) {
        if (int rc = ref_prepare(device)) return rc;
        hipLaunchKernelGGL(sdp_ref_fwd_kernel, dim3(B), dim3(256), ref_lds(M), (hipStream_t)stream, theta, A, state, Vt, lens, N, M, variant == SDP_SW);
        return ref_launched("sdp_ref_fwd_kernel");
    }
    sdp::Params p = {};
    p.sin0 = theta;
    p.sin1 = A;
    p.dout = state;
    p.vout = Vt;
    p.lens = lens;
    p.B = B, p.N = N, p.M = M, p.variant = variant, p.flags = vb.flags;
    if (wants_order(B, N, lens, device)) {
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
        int *order = const_cast<int *>(order_in_state(state, B, N, M, exact));
        hipLaunchKernelGGL(sdp_order_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, lens, order, B, N, M);
        e = hipGetLastError();
        if (e != hipSuccess) return fail_hip(e, "sdp_order_kernel");
        if (B > num_cus(device)) p.order = order;   // (with parts the launch takes it from the state by itself)
    }
    if (!routes_thin(exact, N, M, lens)) return launch(sdp::PASS_FWD, p, device, stream, exact, vb.waves, false, state);
    if (int rc = launch(sdp::PASS_FWD, p, device, stream, false, vb.waves, false, state, 1)) return rc;
    return launch(sdp::PASS_FWD, p, device, stream, true, 0, false, nullptr, 2);
}

int sdp_backward_f32(const float *Et, const float *state, float *E, int B, int N, int M, const int32_t *lens,
                     int variant, int device, void *stream)
{
    if (!Et || !state || !E) return fail(SDP_E_NULLPTR, "sdp_backward_f32: null pointer");
    const VariantBits vb = split_variant(variant);
    const bool exact = exact_for(vb.exact, N, M);
    variant = vb.variant;
    if (int rc = check_shape(B, N, M, variant)) return rc;
    if (vb.ref) {
        if (int rc = ref_prepare(device)) return rc;
        hipLaunchKernelGGL(sdp_ref_bwd_kernel, dim3(B), dim3(256), ref_lds(M), (hipStream_t)stream, Et, state, E, lens, N, M, variant == SDP_SW, vb.et_bcast ? 1 : 0);
        return ref_launched("sdp_ref_bwd_kernel");
    }
    sdp::Params p = {};
    p.vin = Et;
    p.vin_bcast = vb.et_bcast ? 1 : 0;
    p.qin = reinterpret_cast<const uint32_t *>(state);
    p.sout = E;
    p.lens = lens;
    p.B = B, p.N = N, p.M = M, p.variant = variant, p.flags = vb.flags;
    if (lens != nullptr && B > num_cus(device)) p.order = order_in_state(state, B, N, M, exact);
    if (!routes_thin(exact, N, M, lens)) return launch(sdp::PASS_BWD, p, device, stream, exact, vb.waves, false, state);
    if (int rc = launch(sdp::PASS_BWD, p, device, stream, false, vb.waves, false, state, 1)) return rc;
    return launch(sdp::PASS_BWD, p, device, stream, true, 0, false, nullptr, 2);
}

size_t sdp_state_pair_stride(int N, int M, int exact_state)
{
    if (N <= 0 || M <= 0) return 0;
    // the body of the state buffer is B equal records, one per pair: nstrips streams of tpad / 32 units
    // (state_layout); order, bridge rows and dispatch map live behind the LAST pair's record, not between records
    const size_t units = (size_t)sdp::state_nstrips(N) * (sdp::state_tpad(M) / sdp::STATE_UNIT_STEPS);
    return units * (exact_for(exact_state != 0, N, M) ? sdp::STATE2_UNIT_BYTES : sdp::STATEQ_UNIT_BYTES);
}

int sdp_backward_range_f32(const float *Et, const float *state, float *E, int B, int N, int M, int first, int count,
                           int variant, int device, void *stream)
{
    if (!Et || !state || !E) return fail(SDP_E_NULLPTR, "sdp_backward_range_f32: null pointer");
    const VariantBits vb = split_variant(variant);
    const bool exact = exact_for(vb.exact, N, M);
    variant = vb.variant;
    if (int rc = check_shape(B, N, M, variant)) return rc;
    if (first < 0 || count <= 0 || first > B - count) return fail(SDP_E_SHAPE, "sdp_backward_range_f32: [first, first + count) is not inside the batch");
    if (vb.ref) {
        if (int rc = ref_prepare(device)) return rc;
        const size_t plane = (size_t)N * M;
        hipLaunchKernelGGL(sdp_ref_bwd_kernel, dim3(count), dim3(256), ref_lds(M), (hipStream_t)stream, vb.et_bcast ? Et : Et + first,
                           state + (size_t)first * plane * 3, E + (size_t)first * plane, (const int *)nullptr, N, M, variant == SDP_SW, vb.et_bcast ? 1 : 0);
        return ref_launched("sdp_ref_bwd_kernel");
    }
    sdp::Params p = {};
    p.vin = vb.et_bcast ? Et : Et + first;
    p.vin_bcast = vb.et_bcast ? 1 : 0;
    p.qin = reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(state) + (size_t)first * sdp_state_pair_stride(N, M, exact));
    p.sout = E + (size_t)first * N * M;
    p.B = count, p.N = N, p.M = M, p.variant = variant, p.flags = vb.flags;
    // state = nullptr: a launch over part of the batch never spreads pairs over several workgroups -- the bridge rows
    // live behind the record of the batch's LAST pair, and located from a sub-range they would fall into the records of
    // the pairs that follow it
    return launch(sdp::PASS_BWD, p, device, stream, exact, vb.waves, false, nullptr);
}

int sdp_adjoint_forward_f32(const float *state, const float *Ztheta, const float *ZA, float *Vtd, float *state_d,
                            int B, int N, int M, const int32_t *lens, int variant, int device, void *stream)
{
    if (!state || !Ztheta || !Vtd || !state_d) return fail(SDP_E_NULLPTR, "sdp_adjoint_forward_f32: null pointer");
    const VariantBits vb = split_variant(variant);
    variant = vb.variant;
    if (int rc = check_shape(B, N, M, variant)) return rc;
    if (vb.ref) {
        if (int rc = ref_prepare(device)) return rc;
        hipLaunchKernelGGL(sdp_ref_adj_fwd_kernel, dim3(B), dim3(256), ref_lds(M), (hipStream_t)stream, state, Ztheta, ZA, Vtd, state_d, lens, N, M);
        return ref_launched("sdp_ref_adj_fwd_kernel");
    }
    sdp::Params p = {};
    p.qin = reinterpret_cast<const uint32_t *>(state);
    p.sin0 = Ztheta;
    p.sin1 = ZA;
    p.dout = state_d;
    p.vout = Vtd;
    p.lens = lens;
    p.B = B, p.N = N, p.M = M, p.variant = variant, p.flags = vb.flags;
    if (lens != nullptr && B > num_cus(device)) p.order = order_in_state(state, B, N, M, true);
    return launch(sdp::PASS_AFWD, p, device, stream, false, vb.waves);
}

int sdp_adjoint_forward_loss_f32(const float *state, const float *ref, const float *pred, const float *G, const float *scale,
                                 int kind, float *Vtd, float *state_d, int B, int N, int M, const int32_t *lens, int variant,
                                 int device, void *stream)
{
    if (!state || !ref || !pred || !G || !scale || !Vtd || !state_d)
        return fail(SDP_E_NULLPTR, "sdp_adjoint_forward_loss_f32: null pointer");
    if (kind < 0 || kind > 2) return fail(SDP_E_VARIANT, "loss kind must be 0 (cross entropy), 1 (path) or 2 (alignment)");
    const VariantBits vb = split_variant(variant);
    variant = vb.variant;
    if (vb.ref) return fail(SDP_E_VARIANT, "sdp_adjoint_forward_loss_f32: the fused loss seed has no reference-rounding form (use sdp_loss_backward_f32 + sdp_adjoint_forward_f32)");
    if (int rc = check_shape(B, N, M, variant)) return rc;
    sdp::Params p = {};
    p.qin = reinterpret_cast<const uint32_t *>(state);
    p.sin0 = ref;
    p.sin1 = pred;
    p.sin2 = G;
    p.vin = scale;
    p.loss_kind = kind;
    p.dout = state_d;
    p.vout = Vtd;
    p.lens = lens;
    p.B = B, p.N = N, p.M = M, p.variant = variant, p.flags = vb.flags;
    if (lens != nullptr && B > num_cus(device)) p.order = order_in_state(state, B, N, M, true);
    return launch(sdp::PASS_AFWD, p, device, stream, false, vb.waves, true);
}

int sdp_adjoint_backward_f32(const float *E, const float *state, const float *state_d, float *Ed, int B, int N,
                             int M, const int32_t *lens, int variant, int device, void *stream)
{
    if (!E || !state || !state_d || !Ed) return fail(SDP_E_NULLPTR, "sdp_adjoint_backward_f32: null pointer");
    const VariantBits vb = split_variant(variant);
    variant = vb.variant;
    if (int rc = check_shape(B, N, M, variant)) return rc;
    if (vb.ref) {
        if (int rc = ref_prepare(device)) return rc;
        hipLaunchKernelGGL(sdp_ref_adj_bwd_kernel, dim3(B), dim3(256), ref_lds(M), (hipStream_t)stream, E, state, state_d, Ed, lens, N, M);
        return ref_launched("sdp_ref_adj_bwd_kernel");
    }
    sdp::Params p = {};
    p.sin0 = E;
    p.qin = reinterpret_cast<const uint32_t *>(state);
    p.din = reinterpret_cast<const float2 *>(state_d);
    p.sout = Ed;
    p.lens = lens;
    p.B = B, p.N = N, p.M = M, p.variant = variant, p.flags = vb.flags;
    if (lens != nullptr && B > num_cus(device)) p.order = order_in_state(state, B, N, M, true);
    return launch(sdp::PASS_ABWD, p, device, stream, false, vb.waves);
}

// ---- float64 tensors (sdp.h): the reference-arithmetic kernels of sdp_ref.hip with float64 storage ----
size_t sdp_state_bytes_f64(int B, int N, int M)
{
    if (B <= 0 || N <= 0 || M <= 0) return 0;
    return (size_t)B * N * M * 3 * sizeof(double);
}

static int f64_variant(int variant, const char *who, bool &et_bcast)
{
    et_bcast = (variant & SDP_ET_BROADCAST) != 0;
    const int v = variant & ~SDP_ET_BROADCAST;
    if (v != SDP_NW && v != SDP_SW) {
        snprintf(g_err, sizeof(g_err), "%s: variant must be SDP_NW or SDP_SW (float64 tensors take no state / rounding / wave flags)", who);
        return -1;
    }
    return v;
}

int sdp_forward_f64(const double *theta, const double *A, double *state, double *Vt, int B, int N, int M,
                    const int32_t *lens, int variant, int device, void *stream)
{
    if (!theta || !A || !state || !Vt) return fail(SDP_E_NULLPTR, "sdp_forward_f64: null pointer");
    bool bc;
    const int v = f64_variant(variant, "sdp_forward_f64", bc);
    if (v < 0 || bc) return v < 0 ? SDP_E_VARIANT : fail(SDP_E_VARIANT, "sdp_forward_f64: SDP_ET_BROADCAST belongs to the backward sweep");
    if (int rc = check_shape(B, N, M, v)) return rc;
    if (int rc = ref_prepare(device)) return rc;
    hipLaunchKernelGGL(sdp_f64_fwd_kernel, dim3(B), dim3(256), ref_lds(M), (hipStream_t)stream, theta, A, state, Vt, lens, N, M, v == SDP_SW);
    return ref_launched("sdp_f64_fwd_kernel");
}

int sdp_backward_f64(const double *Et, const double *state, double *E, int B, int N, int M,
                     const int32_t *lens, int variant, int device, void *stream)
{
    if (!Et || !state || !E) return fail(SDP_E_NULLPTR, "sdp_backward_f64: null pointer");
    bool bc;
    const int v = f64_variant(variant, "sdp_backward_f64", bc);
    if (v < 0) return SDP_E_VARIANT;
    if (int rc = check_shape(B, N, M, v)) return rc;
    if (int rc = ref_prepare(device)) return rc;
    hipLaunchKernelGGL(sdp_f64_bwd_kernel, dim3(B), dim3(256), ref_lds(M), (hipStream_t)stream, Et, state, E, lens, N, M, v == SDP_SW, bc ? 1 : 0);
    return ref_launched("sdp_f64_bwd_kernel");
}

int sdp_adjoint_forward_f64(const double *state, const double *Ztheta, const double *ZA, double *Vtd, double *state_d,
                            int B, int N, int M, const int32_t *lens, int variant, int device, void *stream)
{
    if (!state || !Ztheta || !Vtd || !state_d) return fail(SDP_E_NULLPTR, "sdp_adjoint_forward_f64: null pointer");
    bool bc;
    const int v = f64_variant(variant, "sdp_adjoint_forward_f64", bc);
    if (v < 0 || bc) return v < 0 ? SDP_E_VARIANT : fail(SDP_E_VARIANT, "sdp_adjoint_forward_f64: SDP_ET_BROADCAST belongs to the backward sweep");
    if (int rc = check_shape(B, N, M, v)) return rc;
    if (int rc = ref_prepare(device)) return rc;
    hipLaunchKernelGGL(sdp_f64_adj_fwd_kernel, dim3(B), dim3(256), ref_lds(M), (hipStream_t)stream, state, Ztheta, ZA, Vtd, state_d, lens, N, M);
    return ref_launched("sdp_f64_adj_fwd_kernel");
}

int sdp_adjoint_backward_f64(const double *E, const double *state, const double *state_d, double *Ed,
                             int B, int N, int M, const int32_t *lens, int variant, int device, void *stream)
{
    if (!E || !state || !state_d || !Ed) return fail(SDP_E_NULLPTR, "sdp_adjoint_backward_f64: null pointer");
    bool bc;
    const int v = f64_variant(variant, "sdp_adjoint_backward_f64", bc);
    if (v < 0 || bc) return v < 0 ? SDP_E_VARIANT : fail(SDP_E_VARIANT, "sdp_adjoint_backward_f64: SDP_ET_BROADCAST belongs to the backward sweep");
    if (int rc = check_shape(B, N, M, v)) return rc;
    if (int rc = ref_prepare(device)) return rc;
    hipLaunchKernelGGL(sdp_f64_adj_bwd_kernel, dim3(B), dim3(256), ref_lds(M), (hipStream_t)stream, E, state, state_d, Ed, lens, N, M);
    return ref_launched("sdp_f64_adj_bwd_kernel");
}

static bool scores_force_f32()
{
#ifdef SDP_EXPERIMENTS
    return (g_dbg.load() & 16) != 0;   // sdp_set_debug(16): the f32-input MFMA kernel for every shape (A/B timing, parity)
#else
    return false;
#endif
}

static bool scores_force_narrow()
{
#ifdef SDP_EXPERIMENTS
    return (g_dbg.load() & 32) != 0;   // sdp_set_debug(32): the 128 x 128 tiles for every shape (A/B timing, bit-identity test)
#else
    return false;
#endif
}
bool scores_unfused()
{
#ifdef SDP_EXPERIMENTS
    return (g_dbg.load() & 2048) != 0;   // sdp_set_debug(2048): backward of the scores with dS in a pass of its own (A/B timing, parity)
#else
    return false;
#endif
}

int sdp_scores_f32(const float *zx, const float *zy, const float *gx, const float *gy, float *theta, float *A, int B, int N,
                   int M, int D, int device, void *stream)
{
    if (!zx || !zy || !theta) return fail(SDP_E_NULLPTR, "sdp_scores_f32: null pointer");
    if ((gx || gy || A) && !(gx && gy && A)) return fail(SDP_E_NULLPTR, "sdp_scores_f32: gx, gy and A go together (all or none)");
    if (B <= 0 || N <= 0 || M <= 0 || D <= 0) return fail(SDP_E_SHAPE, "B, N, M and D must be positive");
    if ((size_t)N * M > ((size_t)1 << 28) || (size_t)N * D > ((size_t)1 << 28) || (size_t)M * D > ((size_t)1 << 28))
        return fail(SDP_E_TOOBIG, "a matrix of one pair exceeds 2^28 elements");
    const long long nz = (long long)(A ? 2 : 1) * B;
    if (nz > 65535) return fail(SDP_E_TOOBIG, "too many pairs for one launch (grid.z)");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
    if (int rc = raise_scores_limits(device)) return rc;
    // whole 16-deep slabs of 16-byte aligned rows: the three-piece bf16 product (sdp_scores.hip); anything else: the
    // f32-input MFMA kernel, which takes ragged D and unaligned rows
    auto aligned16 = [](const void *q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    const bool x6 = D % 16 == 0 && aligned16(zx) && aligned16(zy) && aligned16(gx) && aligned16(gy) && !scores_force_f32();
    const dim3 grid((M + 127) / 128, (N + 127) / 128, (unsigned)nz);
    // 256 x 256 tiles (8 waves, half the cuts and LDS traffic per MFMA) when they do not waste more than a quarter more
    // of the matrix pipe on rows and columns outside the matrices than the 128 x 128 tiles do, and when there are at
    // least two of them per CU (one workgroup per CU: a batch of few large tiles leaves CUs idle -- 3 x 1000 x 701:
    // 129 us against 85 us with the small tiles)
    const long long t128 = (long long)grid.x * grid.y, t256 = (long long)((M + 255) / 256) * ((N + 255) / 256) * 4;
    const bool wide = x6 && !scores_force_narrow() && 4 * t256 <= 5 * t128 && (t256 / 4) * nz >= 2LL * num_cus(device);
    if (wide)
        hipLaunchKernelGGL(sdp_scores_x6w_kernel, dim3((M + 255) / 256, (N + 255) / 256, (unsigned)nz), dim3(512), sdp::SCORES_X6W_LDS_BYTES,
                           (hipStream_t)stream, zx, zy, gx, gy, theta, A, B, N, M, D);
    else if (x6 && t128 * nz <= 2LL * num_cus(device))   // (no CU ever holds a third tile: the build with two workgroups' worth of registers, no scratch)
        hipLaunchKernelGGL(sdp_scores_x6s_kernel, grid, dim3(256), sdp::SCORES_X6_LDS_BYTES, (hipStream_t)stream, zx, zy, gx, gy, theta, A, B, N, M, D);
    else if (x6)
        hipLaunchKernelGGL(sdp_scores_x6_kernel, grid, dim3(256), sdp::SCORES_X6_LDS_BYTES, (hipStream_t)stream, zx, zy, gx, gy, theta, A, B, N, M, D);
    else
        hipLaunchKernelGGL(sdp_scores_kernel, grid, dim3(256), sdp::SCORES_LDS_BYTES, (hipStream_t)stream, zx, zy, gx, gy, theta, A, B, N, M, D);
    e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, x6 ? "sdp_scores_x6_kernel" : "sdp_scores_kernel");
    return 0;
}

size_t sdp_scores_backward_ws_bytes(int B, int N, int M)
{
    if (B <= 0 || N <= 0 || M <= 0) return 0;
    return (size_t)2 * B * N * M * sizeof(float);   // dS of theta, dS of A
}

int sdp_scores_backward_f32(const float *g_theta, const float *g_A, const float *theta, const float *A, const float *zx,
                            const float *zy, const float *gx, const float *gy, float *ws, float *dzx, float *dzy, float *dgx,
                            float *dgy, int B, int N, int M, int D, int device, void *stream)
{
    const bool has_t = g_theta != nullptr, has_a = g_A != nullptr;
    if (!has_t && !has_a) return fail(SDP_E_NULLPTR, "sdp_scores_backward_f32: neither g_theta nor g_A");
    if (!ws || (has_t && !(theta && zx && zy && dzx && dzy)) || (has_a && !(A && gx && gy && dgx && dgy)))
        return fail(SDP_E_NULLPTR, "sdp_scores_backward_f32: null pointer");
    if (B <= 0 || N <= 0 || M <= 0 || D <= 0) return fail(SDP_E_SHAPE, "B, N, M and D must be positive");
    if ((size_t)N * M > ((size_t)1 << 28) || (size_t)N * D > ((size_t)1 << 28) || (size_t)M * D > ((size_t)1 << 28))
        return fail(SDP_E_TOOBIG, "a matrix of one pair exceeds 2^28 elements");
    auto aligned16 = [](const void *q) { return q == nullptr || (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    if (M % 4 != 0 || D % 4 != 0 || !(aligned16(g_theta) && aligned16(g_A) && aligned16(theta) && aligned16(A) && aligned16(zx) && aligned16(zy) &&
                                      aligned16(gx) && aligned16(gy) && aligned16(ws) && aligned16(dzx) && aligned16(dzy) && aligned16(dgx) && aligned16(dgy)))
        return fail(SDP_E_SHAPE, "sdp_scores_backward_f32 needs M and D multiples of 4 and 16-byte aligned tensors");
    const long long nz = (long long)(has_t && has_a ? 2 : 1) * B;
    if (nz > 65535) return fail(SDP_E_TOOBIG, "too many pairs for one launch (grid.z)");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
    if (int rc = raise_scores_limits(device)) return rc;
    const size_t plane = (size_t)B * N * M;
    float *ds_t = ws, *ds_a = ws + plane;
    // tensor 0 / 1 of a launch: theta's operands first when both are asked for
    const float *d0 = has_t ? ds_t : ds_a, *d1 = ds_a;
    const float *y0 = has_t ? zy : gy, *y1 = gy, *x0 = has_t ? zx : gx, *x1 = gx;
    float *cx0 = has_t ? dzx : dgx, *cx1 = dgx, *cy0 = has_t ? dzy : dgy, *cy1 = dgy;
    const dim3 grid_y((D + 255) / 256, (M + 255) / 256, (unsigned)nz), grid_x((D + 255) / 256, (N + 255) / 256, (unsigned)nz);
    if (scores_unfused()) {
        // (experiments build, sdp_set_debug(2048): dS in a pass of its own, then the plain product -- what the fused kernel replaced)
        hipLaunchKernelGGL(sdp_scores_ds_kernel, dim3(num_cus(device) * 8), dim3(256), 0, (hipStream_t)stream, g_theta, g_A, theta, A, ds_t, ds_a, plane / 4);
        e = hipGetLastError();
        if (e != hipSuccess) return fail_hip(e, "sdp_scores_ds_kernel");
        hipLaunchKernelGGL(sdp_scores_bwd_y_kernel, grid_y, dim3(512), sdp::SCORES_X6W_LDS_BYTES, (hipStream_t)stream, d0, d1, x0, x1, cy0, cy1, B, N, M, D);
        e = hipGetLastError();
        if (e != hipSuccess) return fail_hip(e, "sdp_scores_bwd_y_kernel");
    } else {
        // dzy / dgy first: this product reads (g, act) and forms dS on the way into LDS; its first column of tiles writes dS
        // to `ws` for the other product, which follows on the same stream
        const float *g0 = has_t ? g_theta : g_A, *g1 = g_A, *a0 = has_t ? theta : A, *a1 = A;
        const float sg0 = has_t ? -1.0f : 1.0f, sg1 = 1.0f;
        float *w0 = has_t ? ds_t : ds_a, *w1 = ds_a;
        hipLaunchKernelGGL(sdp_scores_bwd_yf_kernel, grid_y, dim3(512), sdp::SCORES_X6W_LDS_BYTES, (hipStream_t)stream, g0, g1, a0, a1, sg0, sg1, x0, x1, w0, w1,
                           cy0, cy1, B, N, M, D);
        e = hipGetLastError();
        if (e != hipSuccess) return fail_hip(e, "sdp_scores_bwd_yf_kernel");
    }
    hipLaunchKernelGGL(sdp_scores_bwd_x_kernel, grid_x, dim3(512), sdp::SCORES_X6W_LDS_BYTES, (hipStream_t)stream, d0, d1, y0, y1, cx0, cx1, B, N, M, D);
    e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "sdp_scores_bwd_x_kernel");
    return 0;
}

int sdp_traceback_capacity(int N, int M) { return (N > 0 && M > 0) ? N + M + 2 : 0; }

int sdp_traceback_rule_i32(const float *grad, int32_t *states, int32_t *counts, int B, int N, int M, const int32_t *lens,
                           int rule, int device, void *stream)
{
    if (!grad || !states || !counts) return fail(SDP_E_NULLPTR, "sdp_traceback_i32: null pointer");
    if (B <= 0 || N <= 0 || M <= 0) return fail(SDP_E_SHAPE, "B, N and M must be positive");
    if (rule != SDP_TRACEBACK_CPU && rule != SDP_TRACEBACK_CUDA) return fail(SDP_E_VARIANT, "traceback rule must be SDP_TRACEBACK_CPU or SDP_TRACEBACK_CUDA");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
    if (rule == SDP_TRACEBACK_CUDA)
        hipLaunchKernelGGL(sdp_traceback_cuda_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, grad, states, counts,
                           lens, B, N, M, sdp_traceback_capacity(N, M));
    else
        hipLaunchKernelGGL(sdp_traceback_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, grad, states, counts,
                           lens, B, N, M, sdp_traceback_capacity(N, M));
    e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "sdp_traceback_kernel");
    return 0;
}

int sdp_traceback_i32(const float *grad, int32_t *states, int32_t *counts, int B, int N, int M, const int32_t *lens,
                      int device, void *stream)
{
    return sdp_traceback_rule_i32(grad, states, counts, B, N, M, lens, SDP_TRACEBACK_CPU, device, stream);
}

int sdp_loss_forward_f32(const float *ref, const float *pred, const float *G, const int32_t *lens, double *acc, int32_t *cnt,
                         int B, int N, int M, int kind, int device, void *stream)
{
    if (!ref || !pred || !G || !acc || !cnt) return fail(SDP_E_NULLPTR, "sdp_loss_forward_f32: null pointer");
    if (B <= 0 || N <= 0 || M <= 0) return fail(SDP_E_SHAPE, "B, N and M must be positive");
    if (kind < 0 || kind > 2) return fail(SDP_E_VARIANT, "loss kind must be 0 (cross entropy), 1 (path) or 2 (alignment)");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
    hipLaunchKernelGGL(sdp_loss_fwd_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, ref, pred, G, lens, acc, cnt, N, M, kind);
    e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "sdp_loss_fwd_kernel");
    return 0;
}

int sdp_loss_backward_f32(const float *ref, const float *pred, const float *G, const int32_t *lens, const float *scale,
                          float *grad, int B, int N, int M, int kind, int device, void *stream)
{
    if (!ref || !pred || !G || !scale || !grad) return fail(SDP_E_NULLPTR, "sdp_loss_backward_f32: null pointer");
    if (B <= 0 || N <= 0 || M <= 0) return fail(SDP_E_SHAPE, "B, N and M must be positive");
    if (kind < 0 || kind > 2) return fail(SDP_E_VARIANT, "loss kind must be 0 (cross entropy), 1 (path) or 2 (alignment)");
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
    const long long groups = (long long)N * ((M + 3) / 4);   // a thread writes four columns per iteration
    int gx = (int)((groups + 256 * 4 - 1) / (256 * 4));
    if (gx < 1) gx = 1;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(sdp_loss_bwd_kernel, dim3(gx, B), dim3(256), 0, (hipStream_t)stream, ref, pred, G, lens, scale, grad, N, M,
                       kind);
    e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "sdp_loss_bwd_kernel");
    return 0;
}


int sdp_selftest(int device)
{
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return fail_hip(e, "hipSetDevice");
    int *d = nullptr;
    e = hipMalloc(&d, 256 * sizeof(int));
    if (e != hipSuccess) return fail_hip(e, "hipMalloc");
    int h[256];
    for (int i = 0; i < 256; ++i) h[i] = 7777;
    (void)hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(sdp_selftest_kernel, dim3(1), dim3(64), 0, 0, d);
    e = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return fail_hip(e, "sdp_selftest_kernel");
    int bad = 0;
    for (int i = 0; i < 64; ++i) bad |= h[i];
    for (int i = 0; i < 64; ++i)
        if (h[64 + i] != 1000 + i) bad |= 128;   // in-range stores must land
    for (int i = 128; i < 192; ++i)
        if (h[i] != 7777) bad |= 256;            // out-of-range stores must be dropped
    if (bad) {
        snprintf(g_err, sizeof(g_err), "sdp_selftest: hardware semantics mismatch, mask 0x%x", bad);
        return SDP_E_SELFTEST;
    }
    return 0;
}

}  // extern "C"
