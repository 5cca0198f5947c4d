// One half step in float64 (gfx950): the arithmetic of the reference's Pool variants, RecModel/wmf_model.py:242-309
// (recompute_factors_par / recompute_factors_bias_par and their *_intern row functions).  With a float64 count matrix those
// keep every row result in float64 -- np.stack of the per-row np.linalg.solve outputs, no cast back to the model dtype --
// so from the second half step on the reference's cores > 1 training runs on float64 factors, float64 Gramians and float64
// row systems.  This file is that pipeline, stated directly and without the whitening of the float32 path:
//   gram64_kernel + gram64_reduce_kernel   G = Y~^T Y~ + lambda I          (:244 / :258; Y~ = Y with column 0 read as 1, :257)
//   solve64_kernel                         per row u: A = G + Y_u^T diag(w) Y_u,  b = Y_u^T (w + 1)   (:285-287 / :305-309),
//                                          w = c_u - bias[idx] for bias models (:279); x = solve(A, b) by LU with partial
//                                          pivoting (np.linalg.solve is LAPACK gesv: the same algorithm, so rows whose
//                                          bias-adjusted weights make A indefinite are solved like the reference solves
//                                          them); rows without stored entries are zero (:274-276 / :296-298)
// It is the correctness path for `cores > 1`, not a fast path: one workgroup per row, the row system in an L2-resident
// workspace slice, f^3 / 3 float64 FMAs per row on the vector units (the reference's own Pool path is ~10x slower than its
// serial loop).
#include "../../include/wmf_hip.h"
#include "wmf_internal.h"

__global__ __launch_bounds__(256) void gram64_kernel(const double* __restrict__ Y, int64_t m, int f, int bias,
                                                     double* __restrict__ partial, int64_t rows_per_block) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(m, r0 + rows_per_block);
    for (int el = threadIdx.x; el < f * f; el += 256) {
        const int i = el / f, j = el % f;
        double acc = 0.0;
        for (int64_t r = r0; r < r1; ++r) {
            const double yi = (bias && i == 0) ? 1.0 : Y[r * f + i];
            const double yj = (bias && j == 0) ? 1.0 : Y[r * f + j];
            acc += yi * yj;
        }
        partial[(int64_t)blockIdx.x * f * f + el] = acc;
    }
}

__global__ __launch_bounds__(256) void gram64_reduce_kernel(const double* __restrict__ partial, int nblocks, int f, double lambda,
                                                            double* __restrict__ G) {
    const int el = blockIdx.x * 256 + threadIdx.x;
    if (el >= f * f) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += partial[(int64_t)b * f * f + el];       // fixed order: reproducible
    G[el] = s + ((el / f == el % f) ? lambda : 0.0);
}

// one workgroup per row (grid-stride); A (f x f), b (f) in this workgroup's workspace slice
__global__ __launch_bounds__(256) void solve64_kernel(const double* __restrict__ Y, int f, int bias, const double* __restrict__ G,
                                                      const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                      const double* __restrict__ vals, int64_t n, double* __restrict__ X,
                                                      double* __restrict__ ws, int32_t* __restrict__ fail) {
    __shared__ double ys[WMF_MAX_F + 3];
    __shared__ double red_v[256];
    __shared__ int red_i[256];
    __shared__ int s_singular;
    const int t = threadIdx.x;
    double* A = ws + (size_t)blockIdx.x * ((size_t)f * f + f);
    double* b = A + (size_t)f * f;
    for (int64_t row = blockIdx.x; row < n; row += gridDim.x) {
        const int64_t lo = indptr[row], hi = indptr[row + 1];
        if (hi == lo) {                                              // no stored entries: zeros (wmf_model.py:274-276, :296-298)
            for (int c = t; c < f; c += 256) X[row * f + c] = 0.0;
            continue;
        }
        for (int el = t; el < f * f; el += 256) A[el] = G[el];
        for (int c = t; c < f; c += 256) b[c] = 0.0;
        if (t == 0) s_singular = 0;
        __syncthreads();
        for (int64_t e = lo; e < hi; ++e) {
            const int64_t idx = indices[e];
            const double w = vals[e] - (bias ? Y[idx * f] : 0.0);    // data - bias[idx], :279
            for (int c = t; c < f; c += 256) ys[c] = (bias && c == 0) ? 1.0 : Y[idx * f + c];
            __syncthreads();
            for (int el = t; el < f * f; el += 256) A[el] += ys[el / f] * (ys[el % f] * w);      // Y_rel.T . (Y_rel * data)
            for (int c = t; c < f; c += 256) b[c] += (w + 1.0) * ys[c];                             // (data + 1) . Y_rel
            __syncthreads();
        }
        // LU with partial pivoting, the right-hand side carried along (gesv: getrf + getrs)
        for (int k = 0; k < f; ++k) {
            double best = -1.0;
            int bi = k;
            for (int i = k + t; i < f; i += 256) {
                const double v = fabs(A[(size_t)i * f + k]);
                if (v > best) { best = v; bi = i; }
            }
            red_v[t] = best; red_i[t] = bi;
            __syncthreads();
            for (int s = 128; s > 0; s >>= 1) {
                if (t < s) {
                    const double ov = red_v[t + s];
                    const int oi = red_i[t + s];
                    if (ov > red_v[t] || (ov == red_v[t] && oi < red_i[t])) { red_v[t] = ov; red_i[t] = oi; }   // first maximum, as idamax
                }
                __syncthreads();
            }
            const int p = red_i[0];
            const double pv = red_v[0];
            if (!(pv > 0.0)) {                                       // exactly singular (or NaN): LinAlgError in the reference
                if (t == 0) s_singular = 1;
                __syncthreads();
                break;
            }
            if (p != k) {
                for (int j = t; j < f; j += 256) {
                    const double a = A[(size_t)k * f + j];
                    A[(size_t)k * f + j] = A[(size_t)p * f + j];
                    A[(size_t)p * f + j] = a;
                }
                if (t == 0) { const double a = b[k]; b[k] = b[p]; b[p] = a; }
            }
            __syncthreads();
            const double piv = A[(size_t)k * f + k];
            for (int i = k + 1 + t; i < f; i += 256) A[(size_t)i * f + k] /= piv;
            __syncthreads();
            const int w_ = f - k - 1;
            for (int el = t; el < w_ * w_; el += 256) {
                const int i = k + 1 + el / w_, j = k + 1 + el % w_;
                A[(size_t)i * f + j] -= A[(size_t)i * f + k] * A[(size_t)k * f + j];
            }
            for (int i = k + 1 + t; i < f; i += 256) b[i] -= A[(size_t)i * f + k] * b[k];
            __syncthreads();
        }
        if (s_singular) {
            if (t == 0) atomicAdd(fail, 1);
            for (int c = t; c < f; c += 256) X[row * f + c] = __builtin_nan("");
            __syncthreads();
            continue;
        }
        for (int k = f - 1; k >= 0; --k) {
            if (t == 0) b[k] /= A[(size_t)k * f + k];
            __syncthreads();
            const double xk = b[k];
            for (int i = t; i < k; i += 256) b[i] -= A[(size_t)i * f + k] * xk;
            __syncthreads();
        }
        for (int c = t; c < f; c += 256) X[row * f + c] = b[c];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void confidence64_kernel(double* __restrict__ v, int64_t n, double alpha, double beta, int mode) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const double x = v[i];
        v[i] = mode == 0 ? alpha * log(1.0 + beta * x) : alpha * x;
    }
}

static int gram64_blocks(int64_t m) {
    int64_t nb = (m + 63) / 64;
    if (nb > 512) nb = 512;
    return (int)(nb < 1 ? 1 : nb);
}
static int solve64_blocks(int64_t n) { return (int)(n < 1 ? 1 : (n > 2048 ? 2048 : n)); }

// workspace: [gram partials nb x f x f][G f x f][solve slices nblocks x (f x f + f)] doubles
int64_t wmf_f64_ws_bytes(int f, int64_t m, int64_t n) {
    const int64_t ff = (int64_t)f * f;
    return 8 * ((int64_t)gram64_blocks(m) * ff + ff + (int64_t)solve64_blocks(n) * (ff + f)) + 256;
}

int wmf_launch_half_step_f64(const double* Y, int64_t m, int f, int bias, const int64_t* indptr, const int32_t* indices,
                             const double* values, int64_t n, double lambda, double* X, void* ws, int32_t* fail, hipStream_t st) {
    const int64_t ff = (int64_t)f * f;
    const int nb = gram64_blocks(m);
    double* partial = static_cast<double*>(ws);
    double* G = partial + (int64_t)nb * ff;
    double* slices = G + ff;
    const int64_t rpb = (m + nb - 1) / nb;
    WMF_LAUNCH("gram64_kernel", gram64_kernel, dim3(nb), dim3(256), 0, st, Y, m, f, bias, partial, rpb);
    WMF_LAUNCH("gram64_reduce_kernel", gram64_reduce_kernel, dim3((unsigned)((ff + 255) / 256)), dim3(256), 0, st, partial, nb, f,
               lambda, G);
    if (n > 0)
        WMF_LAUNCH("solve64_kernel", solve64_kernel, dim3(solve64_blocks(n)), dim3(256), 0, st, Y, f, bias, G, indptr, indices,
                   values, n, X, slices, fail);
    return 0;
}

int wmf_launch_confidence_f64(double* values, int64_t nnz, double alpha, double beta, int mode, hipStream_t st) {
    if (nnz <= 0) return 0;
    int64_t grid = (nnz + 255) / 256;
    if (grid > 4096) grid = 4096;
    WMF_LAUNCH("confidence64_kernel", confidence64_kernel, dim3((unsigned)grid), dim3(256), 0, st, values, nnz, alpha, beta, mode);
    return 0;
}
