/*
 * sdp_oracle.c -- CPU restatement of DeepBLAST's soft-DP alignment recurrences.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle for the HIP engine in
 * deepblast_amd/csrc.  Only tests/, __graft_entry__.smoke() and bench.py's
 * `cpu_baseline` leg may load it; the product path never does.
 *
 * Parity status: PINNED.  oracle/gen_golden.py imports the real reference
 * (/root/reference/deepblast/{nw,sw}.py, numba replaced by the no-op shim in
 * oracle/_shim) and writes tests/golden/ (npz files); tests/test_oracle_golden.py checks
 * every function below against those fixtures, including the reference's own
 * known-answer vectors (deepblast/tests/test_nw.py:43-54, test_sw.py:42-52).
 *
 * What is restated (reference file:line):
 *   soft_max3            deepblast/nw.py:10-27   (_soft_max_numba)
 *   hessian_product3     deepblast/nw.py:30-43   (_soft_max_hessian_product_numba)
 *   forward              deepblast/nw.py:46-62   / sw.py:46-62   (loops from 2)
 *   backward             deepblast/nw.py:120-135 / sw.py:99-114  (loops stop at 2)
 *   adjoint_forward      deepblast/nw.py:178-199 / sw.py:140-161 (same bounds for NW and SW)
 *   adjoint_backward     deepblast/nw.py:251-267 / sw.py:190-209 (same bounds for NW and SW)
 *   batched wrappers     deepblast/nw.py:104-116, 342-355, 357-386 (per-item loop, dtype casts)
 *
 * Numeric types follow the reference exactly: every DP table
 * (V, Q, E, Vd, Qd, Ed) is float64 inside one call; tensors crossing a call
 * boundary are rounded to the storage dtype T (float32 or float64).  Two places
 * in the reference compute in T rather than float64 because numpy promotes
 * T*T -> T:  the Hessian product (np.empty_like(P) with P of dtype T, nw.py:34-41)
 * and the Qd*E products of the adjoint backward (nw.py:261-266).  Both are kept.
 *
 * Index convention: padded coordinates i in [1,N], j in [1,M]; theta/A/ZA are
 * (N,M) row-major 0-based; Q/Qd are (N+2,M+2,3); E/Ztheta/Ed are (N+2,M+2).
 * States x=0 (from (i-1,j)), m=1 (from (i-1,j-1)), y=2 (from (i,j-1))
 * (deepblast/constants.py:1).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SX 0
#define SM 1
#define SY 2

/* nw.py:10-27.  X: 3 maxargs (float64).  Writes softmax weights to P, returns LSE. */
static double soft_max3(const double X[3], double P[3])
{
    double M = X[0];
    for (int i = 1; i < 3; ++i)
        M = X[i] > M ? X[i] : M;
    double S = 0.0;
    for (int i = 0; i < 3; ++i) {
        P[i] = exp(X[i] - M);
        S += P[i];
    }
    for (int i = 0; i < 3; ++i)
        P[i] /= S;
    M += log(S);
    return M;
}

#define ORACLE_IMPL(T, SUF)                                                                   \
                                                                                              \
    /* nw.py:46-62 (variant 0) / sw.py:46-62 (variant 1).  One pair.                   */  \
    /* Q is (N+2,M+2,3) float64 scratch, fully written (zero border, corner = 1).       */  \
    static double forward_one_##SUF(const T *theta, const T *A, int N, int M, int variant,    \
                                    double *V, double *Q)                                     \
    {                                                                                         \
        const int M1 = M + 1, M2 = M + 2;                                                     \
        memset(V, 0, sizeof(double) * (size_t)(N + 1) * M1);                                  \
        memset(Q, 0, sizeof(double) * (size_t)(N + 2) * M2 * 3);                              \
        for (int k = 0; k < 3; ++k)                                                           \
            Q[((size_t)(N + 1) * M2 + (M + 1)) * 3 + k] = 1.0;                                \
        const int lo = variant ? 2 : 1;                                                       \
        double X[3], P[3];                                                                    \
        for (int i = lo; i <= N; ++i) {                                                       \
            for (int j = lo; j <= M; ++j) {                                                   \
                const double a = (double)A[(size_t)(i - 1) * M + (j - 1)];                    \
                X[SX] = a + V[(size_t)(i - 1) * M1 + j];                                      \
                X[SM] = V[(size_t)(i - 1) * M1 + (j - 1)];                                    \
                X[SY] = a + V[(size_t)i * M1 + (j - 1)];                                      \
                const double v = soft_max3(X, P);                                             \
                double *q = Q + ((size_t)i * M2 + j) * 3;                                     \
                q[0] = P[0];                                                                  \
                q[1] = P[1];                                                                  \
                q[2] = P[2];                                                                  \
                V[(size_t)i * M1 + j] = (double)theta[(size_t)(i - 1) * M + (j - 1)] + v;     \
            }                                                                                 \
        }                                                                                     \
        return V[(size_t)N * M1 + M];                                                         \
    }                                                                                         \
                                                                                              \
    /* Batched forward, nw.py:104-116: per-item loop, Vt and Q rounded to T on store.   */  \
    int oracle_forward_##SUF(const T *theta, const T *A, T *Q, T *Vt, int B, int N, int M,    \
                             int variant)                                                     \
    {                                                                                         \
        const size_t nq = (size_t)(N + 2) * (M + 2) * 3;                                      \
        int fail = 0;                                                                         \
        _Pragma("omp parallel")                                                               \
        {                                                                                     \
            double *V = (double *)malloc(sizeof(double) * (size_t)(N + 1) * (M + 1));         \
            double *Qd = (double *)malloc(sizeof(double) * nq);                               \
            if (!V || !Qd) {                                                                  \
                fail = 1;                                                                     \
            } else {                                                                          \
                _Pragma("omp for schedule(dynamic, 1)")                                       \
                for (int b = 0; b < B; ++b) {                                                 \
                    const double vt = forward_one_##SUF(theta + (size_t)b * N * M,            \
                                                        A + (size_t)b * N * M, N, M, variant, \
                                                        V, Qd);                               \
                    Vt[b] = (T)vt;                                                            \
                    T *q = Q + (size_t)b * nq;                                                \
                    for (size_t k = 0; k < nq; ++k)                                           \
                        q[k] = (T)Qd[k];                                                      \
                }                                                                             \
            }                                                                                 \
            free(V);                                                                          \
            free(Qd);                                                                         \
        }                                                                                     \
        return fail ? -1 : 0;                                                                 \
    }                                                                                         \
                                                                                              \
    /* nw.py:120-135 (variant 0) / sw.py:99-114 (variant 1); batched per nw.py:342-355. */  \
    /* Q is the T-rounded tensor saved by forward; E accumulates in float64.            */  \
    int oracle_backward_##SUF(const T *Et, const T *Q, T *E, int B, int N, int M,             \
                              int variant)                                                    \
    {                                                                                         \
        const int M2 = M + 2;                                                                 \
        const size_t ne = (size_t)(N + 2) * M2;                                               \
        int fail = 0;                                                                         \
        _Pragma("omp parallel")                                                               \
        {                                                                                     \
            double *Ew = (double *)malloc(sizeof(double) * ne);                               \
            if (!Ew) {                                                                        \
                fail = 1;                                                                     \
            } else {                                                                          \
                _Pragma("omp for schedule(dynamic, 1)")                                       \
                for (int b = 0; b < B; ++b) {                                                 \
                    const T *q = Q + (size_t)b * ne * 3;                                      \
                    memset(Ew, 0, sizeof(double) * ne);                                       \
                    Ew[(size_t)(N + 1) * M2 + (M + 1)] = (double)Et[b];                       \
                    /* nw.py:126 sets Q[N+1,M+1] = 1 in place; forward already wrote 1. */    \
                    const int stop = variant ? 2 : 1;                                         \
                    for (int i = N; i >= stop; --i) {                                         \
                        for (int j = M; j >= stop; --j) {                                     \
                            const double ex = Ew[(size_t)(i + 1) * M2 + j];                   \
                            const double em = Ew[(size_t)(i + 1) * M2 + (j + 1)];             \
                            const double ey = Ew[(size_t)i * M2 + (j + 1)];                   \
                            const double qx = (double)q[((size_t)(i + 1) * M2 + j) * 3 + SX]; \
                            const double qm =                                                 \
                                (double)q[((size_t)(i + 1) * M2 + (j + 1)) * 3 + SM];         \
                            const double qy = (double)q[((size_t)i * M2 + (j + 1)) * 3 + SY]; \
                            Ew[(size_t)i * M2 + j] = qx * ex + qm * em + qy * ey;             \
                        }                                                                     \
                    }                                                                         \
                    T *e = E + (size_t)b * ne;                                                \
                    for (size_t k = 0; k < ne; ++k)                                           \
                        e[k] = (T)Ew[k];                                                      \
                }                                                                             \
            }                                                                                 \
            free(Ew);                                                                         \
        }                                                                                     \
        return fail ? -1 : 0;                                                                 \
    }                                                                                         \
                                                                                              \
    /* nw.py:178-199 for both variants (sw.py:140-161 keeps the full loop bounds),      */  \
    /* batched per nw.py:381-382.  Hessian product in dtype T (nw.py:30-43, see header). */  \
    int oracle_adjoint_forward_##SUF(const T *Q, const T *Ztheta, const T *ZA, T *Vtd, T *Qd, \
                                     int B, int N, int M)                                     \
    {                                                                                         \
        const int M1 = M + 1, M2 = M + 2;                                                     \
        const size_t ne = (size_t)(N + 2) * M2;                                               \
        int fail = 0;                                                                         \
        _Pragma("omp parallel")                                                               \
        {                                                                                     \
            double *Vd = (double *)malloc(sizeof(double) * (size_t)(N + 1) * M1);             \
            if (!Vd) {                                                                        \
                fail = 1;                                                                     \
            } else {                                                                          \
                _Pragma("omp for schedule(dynamic, 1)")                                       \
                for (int b = 0; b < B; ++b) {                                                 \
                    const T *q = Q + (size_t)b * ne * 3;                                      \
                    const T *zt = Ztheta + (size_t)b * ne;                                    \
                    const T *za = ZA + (size_t)b * N * M;                                     \
                    T *qd = Qd + (size_t)b * ne * 3;                                          \
                    memset(Vd, 0, sizeof(double) * (size_t)(N + 1) * M1);                     \
                    memset(qd, 0, sizeof(T) * ne * 3);                                        \
                    double a[3];                                                              \
                    for (int i = 1; i <= N; ++i) {                                            \
                        for (int j = 1; j <= M; ++j) {                                        \
                            const double z = (double)za[(size_t)(i - 1) * M + (j - 1)];       \
                            a[SX] = z + Vd[(size_t)(i - 1) * M1 + j];                         \
                            a[SM] = Vd[(size_t)(i - 1) * M1 + (j - 1)];                       \
                            a[SY] = z + Vd[(size_t)i * M1 + (j - 1)];                         \
                            const T *p = q + ((size_t)i * M2 + j) * 3;                        \
                            Vd[(size_t)i * M1 + j] = (double)zt[(size_t)i * M2 + j] +         \
                                                     (double)p[SX] * a[0] +                   \
                                                     (double)p[SM] * a[1] +                   \
                                                     (double)p[SY] * a[2];                    \
                            /* prod/res/total live in dtype T (np.empty_like(P)) */           \
                            T prod[3];                                                        \
                            for (int k = 0; k < 3; ++k)                                       \
                                prod[k] = (T)((double)p[k] * a[k]);                           \
                            T total = (T)(prod[0] + prod[1]);                                 \
                            total = (T)(total + prod[2]);                                     \
                            T *r = qd + ((size_t)i * M2 + j) * 3;                             \
                            for (int k = 0; k < 3; ++k) {                                     \
                                const T pt = (T)(p[k] * total);                               \
                                r[k] = (T)(prod[k] - pt);                                     \
                            }                                                                 \
                        }                                                                     \
                    }                                                                         \
                    Vtd[b] = (T)Vd[(size_t)N * M1 + M];                                       \
                }                                                                             \
            }                                                                                 \
            free(Vd);                                                                         \
        }                                                                                     \
        return fail ? -1 : 0;                                                                 \
    }                                                                                         \
                                                                                              \
    /* nw.py:251-267 for both variants (sw.py:190-209), batched per nw.py:383.          */  \
    /* Qd*E products are formed in dtype T (numpy T*T -> T), Q*Ed in float64.           */  \
    int oracle_adjoint_backward_##SUF(const T *E, const T *Q, const T *Qd, T *Ed, int B,      \
                                      int N, int M)                                           \
    {                                                                                         \
        const int M2 = M + 2;                                                                 \
        const size_t ne = (size_t)(N + 2) * M2;                                               \
        int fail = 0;                                                                         \
        _Pragma("omp parallel")                                                               \
        {                                                                                     \
            double *W = (double *)malloc(sizeof(double) * ne);                                \
            if (!W) {                                                                         \
                fail = 1;                                                                     \
            } else {                                                                          \
                _Pragma("omp for schedule(dynamic, 1)")                                       \
                for (int b = 0; b < B; ++b) {                                                 \
                    const T *e = E + (size_t)b * ne;                                          \
                    const T *q = Q + (size_t)b * ne * 3;                                      \
                    const T *qd = Qd + (size_t)b * ne * 3;                                    \
                    memset(W, 0, sizeof(double) * ne);                                        \
                    for (int i = N; i >= 1; --i) {                                            \
                        for (int j = M; j >= 1; --j) {                                        \
                            const size_t sx = (size_t)(i + 1) * M2 + j;                       \
                            const size_t sm = (size_t)(i + 1) * M2 + (j + 1);                 \
                            const size_t sy = (size_t)i * M2 + (j + 1);                       \
                            double acc = (double)(T)(qd[sx * 3 + SX] * e[sx]);                \
                            acc = acc + (double)q[sx * 3 + SX] * W[sx];                       \
                            acc = acc + (double)(T)(qd[sm * 3 + SM] * e[sm]);                 \
                            acc = acc + (double)q[sm * 3 + SM] * W[sm];                       \
                            acc = acc + (double)(T)(qd[sy * 3 + SY] * e[sy]);                 \
                            acc = acc + (double)q[sy * 3 + SY] * W[sy];                       \
                            W[(size_t)i * M2 + j] = acc;                                      \
                        }                                                                     \
                    }                                                                         \
                    T *o = Ed + (size_t)b * ne;                                               \
                    for (size_t k = 0; k < ne; ++k)                                           \
                        o[k] = (T)W[k];                                                       \
                }                                                                             \
            }                                                                                 \
            free(W);                                                                          \
        }                                                                                     \
        return fail ? -1 : 0;                                                                 \
    }

ORACLE_IMPL(float, f32)
ORACLE_IMPL(double, f64)

int oracle_version(void) { return 1; }
