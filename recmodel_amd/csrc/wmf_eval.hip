// Prediction-side kernels (gfx950): WMF.predict over explicit pairs and over the stored entries of
// a CSR utility matrix (RecModel.eval_prec), plus the count -> confidence transform.
// Reference: RecModel/wmf_model.py:191-211 (predict), RecModel/base_model.py:150-179 (eval_prec),
// RecModel/wmf_model.py:119-123 (confidence transform).
#include "wmf_common.h"
#include "wmf_internal.h"

// dot of two factor rows by a 16-lane group: lane gl handles 16-byte pieces gl, gl+16, ...
// bias: column 0 does not enter the product; the score adds both column-0 values instead.
__device__ __forceinline__ float pair_score(const float* __restrict__ xu, const float* __restrict__ yi, int nch, int gl,
                                            int bias) {
    float s = 0.f;
    for (int c = gl; c < nch; c += 16) {
        const float4 a = reinterpret_cast<const float4*>(xu)[c];
        const float4 b = reinterpret_cast<const float4*>(yi)[c];
        float first = a.x * b.x;
        if (bias && c == 0) first = a.x + b.x;
        s += first + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    return wmf_row16_sum(s);
}

__global__ __launch_bounds__(256) void predict_kernel(const float* __restrict__ users, const float* __restrict__ items,
                                                      int ld, int bias, const int32_t* __restrict__ ui, int64_t n_u,
                                                      const int32_t* __restrict__ ii, int64_t n_i, int64_t n,
                                                      float* __restrict__ out) {
    const int gl = threadIdx.x & 15;
    const int nch = ld >> 2;
    for (int64_t p = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); p < n; p += (int64_t)gridDim.x * 16) {
        const int u = ui[n_u == 1 ? 0 : p], i = ii[n_i == 1 ? 0 : p];
        const float s = pair_score(users + (int64_t)u * ld, items + (int64_t)i * ld, nch, gl, bias);
        if (gl == 0) out[p] = s;
    }
}

int wmf_launch_predict(const float* users, const float* items, int f, int ld, int bias, const int32_t* ui, int64_t n_u,
                       const int32_t* ii, int64_t n_i, float* out, hipStream_t st) {
    (void)f;
    const int64_t n = n_u > n_i ? n_u : n_i;
    if (n <= 0) return 0;
    int64_t grid = (n + 15) / 16;
    if (grid > 8192) grid = 8192;
    WmfProfScope ps("predict_kernel", st);
    hipLaunchKernelGGL(predict_kernel, dim3((unsigned)grid), dim3(256), 0, st, users, items, ld, bias, ui, n_u, ii, n_i,
                       n, out);
    return 0;
}

// One wave per user row, four stored entries at a time (one per 16-lane group).  Stored zeros are
// skipped: eval_prec walks utility_mat.nonzero() (base_model.py:163), which drops them.
__global__ __launch_bounds__(256) void eval_kernel(const float* __restrict__ users, const float* __restrict__ items,
                                                   int ld, int bias, const int64_t* __restrict__ indptr,
                                                   const int32_t* __restrict__ indices, const float* __restrict__ vals,
                                                   int64_t n, double* __restrict__ partial) {
    const int lane = threadIdx.x & 63, gl = lane & 15, grp = lane >> 4;
    const int nch = ld >> 2;
    double sq = 0.0, ab = 0.0, cnt = 0.0;
    for (int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); u < n; u += (int64_t)gridDim.x * 4) {
        const int64_t lo = indptr[u], hi = indptr[u + 1];
        const float* xu = users + u * (int64_t)ld;
        for (int64_t jb = lo; jb < hi; jb += 4) {                  // uniform trip count across the 4 groups
            const int64_t j = jb + grp;
            const bool act = j < hi;
            const int64_t e = act ? j : lo;                       // lo < hi inside this loop: a valid entry
            const float v = vals[e];
            const int i = indices[e];
            const float s = pair_score(xu, items + (int64_t)i * ld, nch, gl, bias);
            if (act && gl == 0 && v != 0.f) {
                const double e = (double)v - (double)s;
                sq += e * e; ab += fabs(e); cnt += 1.0;
            }
        }
    }
    // block reduction -> partial[blockIdx][3]
    __shared__ double red[4][3];
    sq = wmf_wave_sum_f64(sq); ab = wmf_wave_sum_f64(ab); cnt = wmf_wave_sum_f64(cnt);
    if (lane == 0) { red[threadIdx.x >> 6][0] = sq; red[threadIdx.x >> 6][1] = ab; red[threadIdx.x >> 6][2] = cnt; }
    __syncthreads();
    if (threadIdx.x < 3) {
        partial[(int64_t)blockIdx.x * 3 + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    }
}

__global__ void eval_finish_kernel(const double* __restrict__ partial, int nblocks, double* __restrict__ out3) {
    const int k = threadIdx.x;
    if (k >= 3) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += partial[(int64_t)b * 3 + k];   // fixed order: reproducible
    out3[k] = s;
}

int wmf_launch_eval(const float* users, const float* items, int f, int ld, int bias, const int64_t* indptr,
                    const int32_t* indices, const float* values, int64_t n, double* out3, double* partial,
                    hipStream_t st) {
    (void)f;
    int64_t grid = (n + 3) / 4;
    if (grid > WMF_EVAL_MAX_BLOCKS) grid = WMF_EVAL_MAX_BLOCKS;
    if (grid < 1) grid = 1;
    WMF_LAUNCH("eval_kernel", eval_kernel, dim3((unsigned)grid), dim3(256), 0, st, users, items, ld, bias, indptr, indices,
               values, n, partial);
    WMF_LAUNCH("eval_finish_kernel", eval_finish_kernel, dim3(1), dim3(64), 0, st, partial, (int)grid, out3);
    return 0;
}

__global__ __launch_bounds__(256) void confidence_kernel(float* __restrict__ v, int64_t n, float alpha, float beta,
                                                         int mode) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float x = v[i];
        v[i] = mode == 0 ? alpha * logf(1.f + beta * x) : alpha * x;
    }
}

int wmf_launch_confidence(float* values, int64_t nnz, double alpha, double beta, int mode, hipStream_t st) {
    if (nnz <= 0) return 0;
    int64_t grid = (nnz + 255) / 256;
    if (grid > 4096) grid = 4096;
    WMF_LAUNCH("confidence_kernel", confidence_kernel, dim3((unsigned)grid), dim3(256), 0, st, values, nnz, (float)alpha,
               (float)beta, mode);
    return 0;
}

// ---- rows of a factor block packed per destination (the need-list exchange): out[i] = in[rows[i]], 16 bytes per lane
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ in, int ld4, const int64_t* __restrict__ rows, int64_t n,
                                                          float* __restrict__ out) {
    const int64_t total = n * ld4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / ld4, c = i - r * ld4;
        reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(in)[rows[r] * ld4 + c];
    }
}

int wmf_launch_gather_rows(const float* in, int ld, const int64_t* rows, int64_t n, float* out, hipStream_t st) {
    if (n <= 0) return 0;
    const int ld4 = ld >> 2;
    int64_t grid = (n * ld4 + 255) / 256;
    if (grid > 16384) grid = 16384;
    WMF_LAUNCH("gather_rows_kernel", gather_rows_kernel, dim3((unsigned)grid), dim3(256), 0, st, in, ld4, rows, n, out);
    return 0;
}
