/* CPU oracle for the WMF/ALS hot path in plain C -- TEST INFRASTRUCTURE ONLY.
 *
 * Restates RecModel/wmf_model.py:213-240 (recompute_factors) and :311-351
 * (recompute_factors_bias) row by row in double precision, the way the reference computes
 * when its count matrix is float64: Gramian of the fixed side
 * (:215 / :332, accumulated here in double), A = G + U^T diag(w) U (:237), b = (w+1)^T U (:239),
 * LU with partial pivoting (LAPACK gesv is what np.linalg.solve calls, :239).
 * Stored zeros contribute, duplicates are not merged, rows without entries give 0.
 * Checked against the golden vectors of the reference in tests/test_oracle_golden.py.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use it.
 *
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <omp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Solve A x = b in place (A is f x f row-major, destroyed; b becomes x).  Returns 0, or k+1 if
 * the k-th pivot is exactly zero. */
static int lu_solve(double* A, double* b, int f) {
    for (int k = 0; k < f; ++k) {
        int p = k;
        double best = fabs(A[k * f + k]);
        for (int i = k + 1; i < f; ++i) {
            double v = fabs(A[i * f + k]);
            if (v > best) { best = v; p = i; }
        }
        if (best == 0.0) return k + 1;
        if (p != k) {
            for (int j = 0; j < f; ++j) { double t = A[k * f + j]; A[k * f + j] = A[p * f + j]; A[p * f + j] = t; }
            double t = b[k]; b[k] = b[p]; b[p] = t;
        }
        double inv = 1.0 / A[k * f + k];
        for (int i = k + 1; i < f; ++i) {
            double l = A[i * f + k] * inv;
            if (l == 0.0) continue;
            for (int j = k + 1; j < f; ++j) A[i * f + j] -= l * A[k * f + j];
            b[i] -= l * b[k];
        }
    }
    for (int k = f - 1; k >= 0; --k) {
        double s = b[k];
        for (int j = k + 1; j < f; ++j) s -= A[k * f + j] * b[j];
        b[k] = s / A[k * f + k];
    }
    return 0;
}

/* X[n, f] (double) = recompute_factors[_bias](Y[m, f] float32, CSR(indptr int64, indices int32,
 * values double), lambda).  bias != 0: column 0 of Y is the fixed side's bias (wmf_model.py:328-343).
 * Returns the number of singular rows (their output is left at 0). */
/* number of OpenMP threads the next calls use; returns the value in effect */
int wmf_oracle_set_threads(int n) {
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
}

int wmf_oracle_half_step(const float* Y, int64_t m, int f, int bias, const int64_t* indptr, const int32_t* indices,
                         const double* values, int64_t n, double lambda, double* X) {
    double* G = (double*)calloc((size_t)f * f, sizeof(double));
    float* Yt = (float*)malloc((size_t)m * f * sizeof(float));
    float* bvec = (float*)calloc((size_t)m, sizeof(float));
    memcpy(Yt, Y, (size_t)m * f * sizeof(float));
    if (bias) for (int64_t r = 0; r < m; ++r) { bvec[r] = Yt[r * f]; Yt[r * f] = 1.0f; }
    /* Gramian: per-thread partial sums over blocks of rows, added in thread order */
#pragma omp parallel
    {
        double* Gt = (double*)calloc((size_t)f * f, sizeof(double));
#pragma omp for schedule(static)
        for (int64_t r = 0; r < m; ++r)
            for (int a = 0; a < f; ++a) {
                double ya = Yt[r * f + a];
                for (int c = 0; c < f; ++c) Gt[a * f + c] += ya * (double)Yt[r * f + c];
            }
#pragma omp for ordered schedule(static, 1)
        for (int t = 0; t < omp_get_num_threads(); ++t) {
#pragma omp ordered
            for (int e = 0; e < f * f; ++e) G[e] += Gt[e];
        }
        free(Gt);
    }
    for (int a = 0; a < f; ++a) G[a * f + a] += lambda;
    int bad = 0;
#pragma omp parallel reduction(+ : bad)
    {
    double* A = (double*)malloc((size_t)f * f * sizeof(double));   /* one scratch matrix per thread */
#pragma omp for schedule(dynamic, 64)
    for (int64_t u = 0; u < n; ++u) {
        double* x = X + u * f;
        memset(x, 0, (size_t)f * sizeof(double));
        int64_t lo = indptr[u], hi = indptr[u + 1];
        if (hi == lo) continue;
        memcpy(A, G, (size_t)f * f * sizeof(double));
        for (int64_t j = lo; j < hi; ++j) {
            const float* y = Yt + (int64_t)indices[j] * f;
            double w = values[j] - (bias ? (double)bvec[indices[j]] : 0.0);
            for (int a = 0; a < f; ++a) {
                double wy = w * (double)y[a];
                x[a] += (w + 1.0) * (double)y[a];
                for (int c = 0; c < f; ++c) A[a * f + c] += wy * (double)y[c];
            }
        }
        if (lu_solve(A, x, f)) { memset(x, 0, (size_t)f * sizeof(double)); bad += 1; }
    }
    free(A);
    }
    free(G); free(Yt); free(bvec);
    return bad;
}
