// A consumer of include/sdp.h that knows nothing of PyTorch: device buffers from hipMalloc, the stream from
// hipStreamCreate, the four sweeps and the traceback through the C ABI -- what the reference-side binding of
// INTEGRATION.md (a cgo / JNI / ctypes stub over plain pointers and sizes) amounts to.  TEST INFRASTRUCTURE: the results
// are checked against the oracle (oracle/sdp_oracle.c, linked here and nowhere in the product); built and run by
// tests/test_c_consumer_gpu.py.  Exit code 0 and "OK" on the last line = every check passed.
#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sdp.h"

extern "C" {
int oracle_forward_f32(const float *theta, const float *A, float *Q, float *Vt, int B, int N, int M, int variant);
int oracle_backward_f32(const float *Et, const float *Q, float *E, int B, int N, int M, int variant);
int oracle_adjoint_forward_f32(const float *Q, const float *Ztheta, const float *ZA, float *Vtd, float *Qd, int B, int N, int M);
int oracle_adjoint_backward_f32(const float *E, const float *Q, const float *Qd, float *Ed, int B, int N, int M);
}

#define HIP_OK(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess) {                                                           \
            std::printf("FAIL %s: %s\n", #call, hipGetErrorString(e_));                   \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)
#define SDP_OK(call)                                                                      \
    do {                                                                                  \
        int rc_ = (call);                                                                 \
        if (rc_ != 0) {                                                                   \
            std::printf("FAIL %s: rc=%d (%s)\n", #call, rc_, sdp_last_error_string());    \
            return 3;                                                                     \
        }                                                                                 \
    } while (0)

static uint64_t g_seed = 0x9E3779B97F4A7C15ull;
static float uni()   // U[0,1), 24 bits
{
    g_seed = g_seed * 6364136223846793005ull + 1442695040888963407ull;
    return (float)((g_seed >> 40) * (1.0 / 16777216.0));
}

template <typename T>
static T *dev(size_t n)
{
    void *p = nullptr;
    return hipMalloc(&p, n * sizeof(T)) == hipSuccess ? static_cast<T *>(p) : nullptr;
}

static double worst(const std::vector<float> &got, const std::vector<float> &want)
{
    double w = 0, scale = 1;
    for (float v : want) scale = std::fmax(scale, std::fabs((double)v));
    for (size_t i = 0; i < got.size(); ++i) w = std::fmax(w, std::fabs((double)got[i] - (double)want[i]));
    return w / scale;
}

int main(int argc, char **argv)
{
    const int variant = (argc > 1 && std::strcmp(argv[1], "sw") == 0) ? SDP_SW : SDP_NW;
    const int B = 6, N = 150, M = 130;
    const size_t plane = (size_t)N * M, pp = (size_t)(N + 2) * (M + 2);
    if (sdp_version() != SDP_VERSION) {
        std::printf("FAIL header / library version %d / %d\n", SDP_VERSION, sdp_version());
        return 1;
    }
    HIP_OK(hipSetDevice(0));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    SDP_OK(sdp_init(0));

    std::vector<float> theta(B * plane), A(B * plane), Z(B * plane), Et(B);
    for (auto &v : theta) v = uni();
    for (auto &v : A) v = -uni();
    for (auto &v : Z) v = 2.f * uni() - 1.f;
    for (auto &v : Et) v = 0.5f + uni();

    float *d_theta = dev<float>(B * plane), *d_A = dev<float>(B * plane), *d_Z = dev<float>(B * plane), *d_Et = dev<float>(B);
    float *d_E = dev<float>(B * plane), *d_Ed = dev<float>(B * plane), *d_Vt = dev<float>(B), *d_Vtd = dev<float>(B);
    const size_t sbytes = sdp_state_bytes_v(B, N, M, variant | SDP_EXACT_STATE), dbytes = sdp_state_d_bytes_v(B, N, M, variant);
    float *d_state = dev<float>(sbytes / 4 + 1), *d_state_d = dev<float>(dbytes / 4 + 1);
    const int cap = sdp_traceback_capacity(N, M);
    int32_t *d_states = dev<int32_t>((size_t)B * cap * 3), *d_counts = dev<int32_t>(B);
    if (!d_theta || !d_A || !d_Z || !d_Et || !d_E || !d_Ed || !d_Vt || !d_Vtd || !d_state || !d_state_d || !d_states || !d_counts) {
        std::printf("FAIL hipMalloc\n");
        return 2;
    }
    HIP_OK(hipMemcpyAsync(d_theta, theta.data(), B * plane * 4, hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(d_A, A.data(), B * plane * 4, hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(d_Z, Z.data(), B * plane * 4, hipMemcpyHostToDevice, stream));
    HIP_OK(hipMemcpyAsync(d_Et, Et.data(), B * 4, hipMemcpyHostToDevice, stream));

    // error behaviour first: a null pointer and a bad shape are refused with a message, nothing is launched
    if (sdp_forward_f32(nullptr, d_A, d_state, d_Vt, B, N, M, nullptr, variant, 0, stream) != SDP_E_NULLPTR || !*sdp_last_error_string()) {
        std::printf("FAIL null pointer not refused\n");
        return 4;
    }
    if (sdp_forward_f32(d_theta, d_A, d_state, d_Vt, B, N, sdp_max_cols() + 1, nullptr, variant, 0, stream) != SDP_E_MAXCOLS) {
        std::printf("FAIL M > sdp_max_cols() not refused\n");
        return 4;
    }

    // the four sweeps on one exact state (what decode() + a loss on the alignment matrix asks for), then the walks
    SDP_OK(sdp_forward_f32(d_theta, d_A, d_state, d_Vt, B, N, M, nullptr, variant | SDP_EXACT_STATE, 0, stream));
    SDP_OK(sdp_backward_f32(d_Et, d_state, d_E, B, N, M, nullptr, variant | SDP_EXACT_STATE, 0, stream));
    SDP_OK(sdp_adjoint_forward_f32(d_state, d_Z, nullptr, d_Vtd, d_state_d, B, N, M, nullptr, variant, 0, stream));
    SDP_OK(sdp_adjoint_backward_f32(d_E, d_state, d_state_d, d_Ed, B, N, M, nullptr, variant, 0, stream));
    SDP_OK(sdp_traceback_i32(d_E, d_states, d_counts, B, N, M, nullptr, 0, stream));
    std::vector<float> E(B * plane), Ed(B * plane), Vt(B), Vtd(B);
    std::vector<int32_t> counts(B), states((size_t)B * cap * 3);
    HIP_OK(hipMemcpyAsync(E.data(), d_E, B * plane * 4, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(Ed.data(), d_Ed, B * plane * 4, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(Vt.data(), d_Vt, B * 4, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(Vtd.data(), d_Vtd, B * 4, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(counts.data(), d_counts, B * 4, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(states.data(), d_states, states.size() * 4, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    int32_t info[4] = {0, 0, 0, 0};
    SDP_OK(sdp_device_status(0, info));
    if (info[0] != 0) {
        std::printf("FAIL a strip hand-off timed out (%d)\n", info[0]);
        return 5;
    }

    // the oracle on the host (padded layouts of the reference: Q (B,N+2,M+2,3), E (B,N+2,M+2))
    std::vector<float> Q(B * pp * 3), Qd(B * pp * 3), Ep(B * pp), Edp(B * pp), Zp(B * pp, 0.f), ZA(B * plane, 0.f), rVt(B), rVtd(B);
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < M; ++j) Zp[b * pp + (size_t)(i + 1) * (M + 2) + (j + 1)] = Z[b * plane + (size_t)i * M + j];
    if (oracle_forward_f32(theta.data(), A.data(), Q.data(), rVt.data(), B, N, M, variant) ||
        oracle_backward_f32(Et.data(), Q.data(), Ep.data(), B, N, M, variant) ||
        oracle_adjoint_forward_f32(Q.data(), Zp.data(), ZA.data(), rVtd.data(), Qd.data(), B, N, M) ||
        oracle_adjoint_backward_f32(Ep.data(), Q.data(), Qd.data(), Edp.data(), B, N, M)) {
        std::printf("FAIL oracle\n");
        return 6;
    }
    std::vector<float> rE(B * plane), rEd(B * plane);
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < N; ++i)
            for (int j = 0; j < M; ++j) {
                rE[b * plane + (size_t)i * M + j] = Ep[b * pp + (size_t)(i + 1) * (M + 2) + (j + 1)];
                rEd[b * plane + (size_t)i * M + j] = Edp[b * pp + (size_t)(i + 1) * (M + 2) + (j + 1)];
            }
    const double eVt = worst(Vt, rVt), eE = worst(E, rE), eVtd = worst(Vtd, rVtd), eEd = worst(Ed, rEd);
    std::printf("%s %dx%dx%d: Vt %.2e  E %.2e  Vtd %.2e  Ed %.2e (bound 1e-4)\n", variant == SDP_SW ? "sw" : "nw", B, N, M, eVt, eE, eVtd, eEd);
    if (!(eVt <= 1e-4 && eE <= 1e-4 && eVtd <= 1e-4 && eEd <= 1e-4)) {
        std::printf("FAIL parity\n");
        return 7;
    }
    // the walks: each ends at the last cell in state m (1); Needleman-Wunsch walks start at (0, 0) and move by one of the three
    // steps (the Smith-Waterman decoder's walk keeps the reference's index wrap at the border -- sw.py:328-371 -- and is
    // compared entry for entry with the host walk in the Python suite)
    for (int b = 0; b < B; ++b) {
        const int n = counts[b];
        const int32_t *s = states.data() + (size_t)b * cap * 3;
        bool ok = n >= 1 && n <= cap && s[3 * (n - 1)] == N - 1 && s[3 * (n - 1) + 1] == M - 1 && s[3 * (n - 1) + 2] == 1;
        if (variant == SDP_NW) ok = ok && n >= (N > M ? N : M) && s[0] == 0 && s[1] == 0;
        for (int k = 1; ok && variant == SDP_NW && k < n; ++k) {
            const int di = s[3 * k] - s[3 * (k - 1)], dj = s[3 * k + 1] - s[3 * (k - 1) + 1];
            ok = (di == 1 && dj == 0) || (di == 1 && dj == 1) || (di == 0 && dj == 1);
        }
        if (!ok) {
            std::printf("FAIL walk of pair %d (%d states)\n", b, n);
            return 8;
        }
    }
    hipFree(d_theta), hipFree(d_A), hipFree(d_Z), hipFree(d_Et), hipFree(d_E), hipFree(d_Ed), hipFree(d_Vt), hipFree(d_Vtd);
    hipFree(d_state), hipFree(d_state_d), hipFree(d_states), hipFree(d_counts);
    hipStreamDestroy(stream);
    std::printf("OK\n");
    return 0;
}
