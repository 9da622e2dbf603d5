// sdp_ref.hip -- the four sweeps in the REFERENCE's arithmetic (variant | SDP_REF_ROUNDING), for gfx950.
//
// The fast sweeps (sdp_kernels.hip) keep every product of the second-order pair in float64 and therefore sit on the
// float64 answer.  The reference does not: numpy forms the soft-max Hessian product and the Qd * E products in the
// STORAGE dtype (deepblast/nw.py:30-43 `np.empty_like(P)`, nw.py:261-266 float32 * float32), and it rounds Q to fp32
// once, from a float64 exp / log / division (nw.py:10-27, 115).  On long saturated alignments those fp32 roundings are a
// random walk of ~4e-6 per soft cell at |Vd| ~ 60 and move Ed by 1-2e-4 against the float64 result (DESIGN.md section 2,
// "Where the fp32 reference is the noisy one") -- a corner in which "within 1e-4 of the reference" can only be met by
// making the same roundings from the same weights.  That is what these kernels do, operation for operation:
//
//   forward   nw.py:46-62 / sw.py:46-62     V, exp, log, division in float64; Q rounded to fp32 once, ALL THREE weights kept
//   backward  nw.py:120-135 / sw.py:99-114   E accumulated in float64 from the fp32 weights
//   adjoint forward   nw.py:178-199          Vd in float64; prod, total, res of the Hessian product rounded to fp32 where
//                                            numpy rounds them (nw.py:30-43)
//   adjoint backward  nw.py:251-267          Qd * E products rounded to fp32, Q * Ed in float64, the reference's order of terms
//
// States are the reference's own (minus the zero border): Q and Qd as (B, N, M, 3) fp32, 12 bytes per cell.
//
// The same kernels instantiated for float64 STORAGE are the sdp_*_f64 entry points: the reference's CPU classes take
// float64 tensors as they come (its own tests run gradcheck / gradgradcheck and the decoding test on them,
// deepblast/tests/test_nw.py:46-90), and with float64 storage every "rounded to the storage dtype" above is no rounding
// at all -- the sweeps are the float64 recurrences, 24 bytes of state per cell.
// Schedule: one workgroup per pair, a barrier per anti-diagonal, three rolling diagonals of float64 in LDS.  Nothing
// here is tuned -- a 256 x 512 x 512 batch takes milliseconds, not the fast path's 0.2 ms; this mode exists to be
// compared against, and for callers who need the reference's numbers rather than the more accurate ones.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sdp_kernels.h"

namespace {

constexpr int RT = 256;   // threads per pair

__device__ __forceinline__ void pair_dims(const int *lens, int b, int N, int M, int &n, int &m)
{
    n = N, m = M;
    if (lens) {
        n = lens[2 * b], m = lens[2 * b + 1];
        n = n < 1 ? 1 : (n > N ? N : n);
        m = m < 1 ? 1 : (m > M ? M : m);
    }
}

// E / Ed are dense (B, N, M): zero outside the pair's n x m block (lengths-aware mode)
template <typename T>
__device__ void zero_outside(T *out, int n, int m, int N, int M)
{
    if (n == N && m == M) return;
    for (size_t e = (size_t)n * M + threadIdx.x; e < (size_t)N * M; e += RT) out[e] = (T)0;
    if (m < M)
        for (int r = 0; r < n; ++r)
            for (int c = m + (int)threadIdx.x; c < M; c += RT) out[(size_t)r * M + c] = (T)0;
}

// deepblast/nw.py:10-27 + 46-62 (sw.py:46-62: loops from 2, row 1 / column 1 keep V = 0, Q = 0)
template <typename T>
__device__ __forceinline__ void ref_fwd(const T *theta, const T *A, T *Q, T *Vt, const int *lens, int N, int M, int sw)
{
    extern __shared__ double sh[];
    const int b = blockIdx.x, P = M + 2;
    int n, m;
    pair_dims(lens, b, N, M, n, m);
    const size_t plane = (size_t)N * M;
    const T *th = theta + b * plane, *ga = A + b * plane;
    T *q = Q + b * plane * 3;
    for (int d = 0; d <= n + m; ++d) {   // anti-diagonal i + j = d over the padded table V[0..n][0..m]
        double *cur = sh + (d % 3) * P;
        const double *p1 = sh + ((d + 2) % 3) * P, *p2 = sh + ((d + 1) % 3) * P;   // diagonals d - 1, d - 2, indexed by j
        const int jlo = d - n > 0 ? d - n : 0, jhi = d < m ? d : m;
        for (int j = jlo + (int)threadIdx.x; j <= jhi; j += RT) {
            const int i = d - j;
            double v = 0.0;
            if (i >= 1 && j >= 1) {
                T *qc = q + ((size_t)(i - 1) * M + (j - 1)) * 3;
                if (sw && (i == 1 || j == 1)) {
                    qc[0] = qc[1] = qc[2] = (T)0;
                } else {
                    const double a = (double)ga[(size_t)(i - 1) * M + (j - 1)];
                    double X[3], Pw[3];
                    X[0] = a + p1[j];        // x: V[i-1, j]
                    X[1] = p2[j - 1];        // m: V[i-1, j-1]
                    X[2] = a + p1[j - 1];    // y: V[i, j-1]
                    double mx = X[0];
                    for (int k = 1; k < 3; ++k) mx = X[k] > mx ? X[k] : mx;
                    double S = 0.0;
                    for (int k = 0; k < 3; ++k) {
                        Pw[k] = exp(X[k] - mx);
                        S += Pw[k];
                    }
                    for (int k = 0; k < 3; ++k) qc[k] = (T)(Pw[k] / S);
                    v = (double)th[(size_t)(i - 1) * M + (j - 1)] + (mx + log(S));
                }
            }
            cur[j] = v;
            if (i == n && j == m) Vt[b] = (T)v;
        }
        __syncthreads();
    }
}

// deepblast/nw.py:120-135 (sw.py:99-114: loops stop at 2)
template <typename T>
__device__ __forceinline__ void ref_bwd(const T *Et, const T *Q, T *E, const int *lens, int N, int M, int sw, int et_bcast)
{
    extern __shared__ double sh[];
    const int b = blockIdx.x, P = M + 2;
    int n, m;
    pair_dims(lens, b, N, M, n, m);
    const size_t plane = (size_t)N * M;
    const T *q = Q + b * plane * 3;
    T *e = E + b * plane;
    const double et = (double)Et[et_bcast ? 0 : b];
    const int stop = sw ? 2 : 1;
    for (int d = n + m; d >= 2; --d) {
        double *cur = sh + (d % 3) * P;
        const double *n1 = sh + ((d + 1) % 3) * P, *n2 = sh + ((d + 2) % 3) * P;   // diagonals d + 1, d + 2, indexed by j
        const int jlo = d - n > 1 ? d - n : 1, jhi = d - 1 < m ? d - 1 : m;
        for (int j = jlo + (int)threadIdx.x; j <= jhi; j += RT) {
            const int i = d - j;
            double v = 0.0;
            if (i >= stop && j >= stop) {
                double ex = 0.0, em = 0.0, ey = 0.0, qx = 0.0, qm = 0.0, qy = 0.0;
                if (i + 1 <= n) ex = n1[j], qx = (double)q[((size_t)i * M + (j - 1)) * 3 + 0];                     // Q[i+1, j, x] E[i+1, j]
                if (i + 1 <= n && j + 1 <= m) em = n2[j + 1], qm = (double)q[((size_t)i * M + j) * 3 + 1];       // Q[i+1, j+1, m] E[i+1, j+1]
                else if (i == n && j == m) em = et, qm = 1.0;                                                    // the corner: Q[N+1, M+1] = 1, E[N+1, M+1] = Et
                if (j + 1 <= m) ey = n1[j + 1], qy = (double)q[((size_t)(i - 1) * M + j) * 3 + 2];               // Q[i, j+1, y] E[i, j+1]
                v = qx * ex + qm * em + qy * ey;
            }
            cur[j] = v;
            e[(size_t)(i - 1) * M + (j - 1)] = (T)v;
        }
        __syncthreads();
    }
    zero_outside(e, n, m, N, M);
}

// deepblast/nw.py:178-199 with the Hessian product of nw.py:30-43 in the storage dtype (both variants: sw.py:140-161 keeps
// the full loop bounds; Q is zero on its row 1 / column 1)
template <typename T>
__device__ __forceinline__ void ref_adj_fwd(const T *Q, const T *Ztheta, const T *ZA, T *Vtd, T *Qd, const int *lens, int N, int M)
{
    extern __shared__ double sh[];
    const int b = blockIdx.x, P = M + 2;
    int n, m;
    pair_dims(lens, b, N, M, n, m);
    const size_t plane = (size_t)N * M;
    const T *q = Q + b * plane * 3, *zt = Ztheta + b * plane, *za = ZA ? ZA + b * plane : nullptr;
    T *qd = Qd + b * plane * 3;
    for (int d = 0; d <= n + m; ++d) {
        double *cur = sh + (d % 3) * P;
        const double *p1 = sh + ((d + 2) % 3) * P, *p2 = sh + ((d + 1) % 3) * P;
        const int jlo = d - n > 0 ? d - n : 0, jhi = d < m ? d : m;
        for (int j = jlo + (int)threadIdx.x; j <= jhi; j += RT) {
            const int i = d - j;
            double v = 0.0;
            if (i >= 1 && j >= 1) {
                const size_t c = (size_t)(i - 1) * M + (j - 1);
                const double z = za ? (double)za[c] : 0.0;
                double a[3];
                a[0] = z + p1[j];
                a[1] = p2[j - 1];
                a[2] = z + p1[j - 1];
                const T *p = q + c * 3;
                v = (double)zt[c] + (double)p[0] * a[0] + (double)p[1] * a[1] + (double)p[2] * a[2];
                T prod[3];   // prod / total / res live in the storage dtype (np.empty_like(P))
                for (int k = 0; k < 3; ++k) prod[k] = (T)((double)p[k] * a[k]);
                T total = prod[0] + prod[1];
                total = total + prod[2];
                for (int k = 0; k < 3; ++k) {
                    const T pt = p[k] * total;
                    qd[c * 3 + k] = prod[k] - pt;
                }
            }
            cur[j] = v;
            if (i == n && j == m) Vtd[b] = (T)v;
        }
        __syncthreads();
    }
}

// deepblast/nw.py:251-267: Qd * E in the storage dtype, Q * Ed in float64, the reference's order of terms
template <typename T>
__device__ __forceinline__ void ref_adj_bwd(const T *E, const T *Q, const T *Qd, T *Ed, const int *lens, int N, int M)
{
    extern __shared__ double sh[];
    const int b = blockIdx.x, P = M + 2;
    int n, m;
    pair_dims(lens, b, N, M, n, m);
    const size_t plane = (size_t)N * M;
    const T *e = E + b * plane, *q = Q + b * plane * 3, *qd = Qd + b * plane * 3;
    T *o = Ed + b * plane;
    for (int d = n + m; d >= 2; --d) {
        double *cur = sh + (d % 3) * P;
        const double *n1 = sh + ((d + 1) % 3) * P, *n2 = sh + ((d + 2) % 3) * P;
        const int jlo = d - n > 1 ? d - n : 1, jhi = d - 1 < m ? d - 1 : m;
        for (int j = jlo + (int)threadIdx.x; j <= jhi; j += RT) {
            const int i = d - j;
            double acc = 0.0;
            if (i + 1 <= n) {                      // source (i+1, j), state x
                const size_t s = (size_t)i * M + (j - 1);
                const T t = qd[s * 3 + 0] * e[s];
                acc = (double)t;
                acc = acc + (double)q[s * 3 + 0] * n1[j];
            }
            if (i + 1 <= n && j + 1 <= m) {        // source (i+1, j+1), state m
                const size_t s = (size_t)i * M + j;
                const T t = qd[s * 3 + 1] * e[s];
                acc = acc + (double)t;
                acc = acc + (double)q[s * 3 + 1] * n2[j + 1];
            }
            if (j + 1 <= m) {                      // source (i, j+1), state y
                const size_t s = (size_t)(i - 1) * M + j;
                const T t = qd[s * 3 + 2] * e[s];
                acc = acc + (double)t;
                acc = acc + (double)q[s * 3 + 2] * n1[j + 1];
            }
            cur[j] = acc;
            o[(size_t)(i - 1) * M + (j - 1)] = (T)acc;
        }
        __syncthreads();
    }
    zero_outside(o, n, m, N, M);
}

}  // namespace

// ---- the kernels: float32 storage (variant | SDP_REF_ROUNDING) and float64 storage (sdp_*_f64) ----
extern "C" __global__ void __launch_bounds__(256) sdp_ref_fwd_kernel(const float *theta, const float *A, float *Q, float *Vt, const int *lens,
                                                                     int N, int M, int sw)
{
    ref_fwd<float>(theta, A, Q, Vt, lens, N, M, sw);
}
extern "C" __global__ void __launch_bounds__(256) sdp_ref_bwd_kernel(const float *Et, const float *Q, float *E, const int *lens, int N, int M,
                                                                     int sw, int et_bcast)
{
    ref_bwd<float>(Et, Q, E, lens, N, M, sw, et_bcast);
}
extern "C" __global__ void __launch_bounds__(256) sdp_ref_adj_fwd_kernel(const float *Q, const float *Ztheta, const float *ZA, float *Vtd, float *Qd,
                                                                         const int *lens, int N, int M)
{
    ref_adj_fwd<float>(Q, Ztheta, ZA, Vtd, Qd, lens, N, M);
}
extern "C" __global__ void __launch_bounds__(256) sdp_ref_adj_bwd_kernel(const float *E, const float *Q, const float *Qd, float *Ed, const int *lens,
                                                                         int N, int M)
{
    ref_adj_bwd<float>(E, Q, Qd, Ed, lens, N, M);
}
extern "C" __global__ void __launch_bounds__(256) sdp_f64_fwd_kernel(const double *theta, const double *A, double *Q, double *Vt, const int *lens,
                                                                     int N, int M, int sw)
{
    ref_fwd<double>(theta, A, Q, Vt, lens, N, M, sw);
}
extern "C" __global__ void __launch_bounds__(256) sdp_f64_bwd_kernel(const double *Et, const double *Q, double *E, const int *lens, int N, int M,
                                                                     int sw, int et_bcast)
{
    ref_bwd<double>(Et, Q, E, lens, N, M, sw, et_bcast);
}
extern "C" __global__ void __launch_bounds__(256) sdp_f64_adj_fwd_kernel(const double *Q, const double *Ztheta, const double *ZA, double *Vtd, double *Qd,
                                                                         const int *lens, int N, int M)
{
    ref_adj_fwd<double>(Q, Ztheta, ZA, Vtd, Qd, lens, N, M);
}
extern "C" __global__ void __launch_bounds__(256) sdp_f64_adj_bwd_kernel(const double *E, const double *Q, const double *Qd, double *Ed, const int *lens,
                                                                         int N, int M)
{
    ref_adj_bwd<double>(E, Q, Qd, Ed, lens, N, M);
}
