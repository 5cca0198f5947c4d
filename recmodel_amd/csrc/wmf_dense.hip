// Dense shared-work kernels of one ALS half step (gfx950):
//   gram       G_sum = Y~^T Y~                      f32 MFMA, fp64 cross-wave reduction
//   factorize  G_sum + lambda I = L L^T (fp64), W_white = L^-T, W_unwhite = L^-1 (fp32 out)
//   transform  out = in~ . W                        f32 MFMA, W staged in LDS
// Reference arithmetic: RecModel/wmf_model.py:215 and :328-332 (Gramian of the fixed side, bias
// column forced to one), and the shared part of the per-row solve (:239, :350).
#include "wmf_common.h"
#include "wmf_internal.h"

#include <utility>

// ------------------------------------------------------------------------------------------ gram
// One wave per workgroup.  A wave walks a contiguous range of 4-row steps; at each step lane
// (r = l & 15, q = l >> 4) loads Y~[4s + q][16 fb + r] for every 16-column feature block fb.  The
// same registers are both MFMA operands: tile(bi, bj) += frag[bi]^T frag[bj].  Tiles with
// bi <= bj only.  Each wave writes its partial tiles to `partial`; gram_reduce sums them in fp64.
template <int NFB, int NSPLIT, int S>
__device__ __forceinline__ void gram_body(const float* __restrict__ Y, int64_t m, int f, int ld, int bias,
                                          float* __restrict__ partial, int64_t step_lo, int64_t step_hi) {
    constexpr int NT = NFB * (NFB + 1) / 2;
    constexpr int NACC = (NT + NSPLIT - 1) / NSPLIT;
    const int lane = threadIdx.x & 63;                           // (NSPLIT = 4: the four waves of a workgroup, one tile quarter each)
    const int r = lane & 15, q = lane >> 4;
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    // U steps (4 rows each) per trip; the loads of the next trip are issued before this trip's MFMAs and
    // stay untouched until then (raw loads: masks and the bias column are applied at use).
    constexpr int U = 4;
    float cur[U][NFB], nxt[U][NFB];
    const int last_col = min(16 * (NFB - 1) + r, ld - 1);
    const float col_mask_last = (16 * (NFB - 1) + r < f) ? 1.f : 0.f;
    auto load_trip = [&](int64_t s0, float (&fr)[U][NFB]) {
#pragma unroll
        for (int uu = 0; uu < U; ++uu) {
            const int64_t row = min(4 * (s0 + uu) + q, m - 1);                        // clamped: masked at use
            const float* yrow = Y + row * (int64_t)ld;
#pragma unroll
            for (int fb = 0; fb < NFB - 1; ++fb) fr[uu][fb] = yrow[16 * fb + r];
            fr[uu][NFB - 1] = yrow[last_col];
        }
    };
    if (step_lo < step_hi) load_trip(step_lo, cur);
    for (int64_t s0 = step_lo; s0 < step_hi; s0 += U) {
        if (s0 + U < step_hi) load_trip(s0 + U, nxt);
#pragma unroll
        for (int uu = 0; uu < U; ++uu) {
            const int64_t s = s0 + uu;
            const float rmask = (s < step_hi && 4 * s + q < m) ? 1.f : 0.f;
            float frag[NFB];
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) frag[fb] = cur[uu][fb] * rmask;
            frag[NFB - 1] *= col_mask_last;
            if (bias && r == 0) frag[0] = rmask;                                     // column 0 reads as 1 (wmf_model.py:331)
            int t = 0;
#pragma unroll
            for (int bi = 0; bi < NFB; ++bi) {
#pragma unroll
                for (int bj = bi; bj < NFB; ++bj, ++t) {
                    if (t % NSPLIT == S) acc[t / NSPLIT] = WMF_MFMA16(frag[bi], frag[bj], acc[t / NSPLIT]);
                }
            }
        }
#pragma unroll
        for (int uu = 0; uu < U; ++uu)
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) cur[uu][fb] = nxt[uu][fb];
    }
    // partial layout: [wave][tile][reg][lane]
    float* out = partial + (int64_t)blockIdx.x * NT * 256;
    int t = 0;
#pragma unroll
    for (int bi = 0; bi < NFB; ++bi) {
#pragma unroll
        for (int bj = bi; bj < NFB; ++bj, ++t) {
            if (t % NSPLIT == S) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) out[(t * 4 + reg) * 64 + lane] = acc[t / NSPLIT][reg];
            }
        }
    }
}

// The same Gramian with split-bf16 products (wide factors, where the f32 MFMAs of gram_body are what bounds the kernel:
// 1.6 ms for 11 M rows at f = 129): a chunk of 32 rows per trip, lane (r, q) loads Y~[32 c + 8 q + j][16 fb + r], j = 0..7
// -- the K index of a 16x16x32 MFMA -- splits every value into three bf16 parts (exact) and issues six bf16 MFMAs per tile
// (lo.hi, mid.mid, hi.lo, mid.hi, hi.mid, hi.hi; the dropped products are below 2^-24 of the term): the accuracy of the
// f32 path at 6 x 16 instead of 8 x 32 MFMA cycles per tile and 32 rows.
typedef __bf16 gram_bf16x8 __attribute__((ext_vector_type(8)));
template <int NFB>
__device__ __forceinline__ void gram_body6(const float* __restrict__ Y, int64_t m, int f, int ld, int bias,
                                           float* __restrict__ partial, int64_t step_lo, int64_t step_hi) {
    constexpr int NT = NFB * (NFB + 1) / 2;
    const int lane = threadIdx.x;
    const int r = lane & 15, q = lane >> 4;
    f32x4 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int last_col = min(16 * (NFB - 1) + r, ld - 1);
    const float col_mask_last = (16 * (NFB - 1) + r < f) ? 1.f : 0.f;
    // chunks of 32 rows dealt round-robin over the waves (step_lo = this wave, step_hi = the number of waves): at any moment the
    // resident waves read one contiguous stretch of the matrix
    const int64_t row_lo = 32 * step_lo, row_hi = m, stride = 32 * step_hi;
    float cur[8][NFB], nxt[8][NFB];
    auto load_chunk = [&](int64_t c0, float (&fr)[8][NFB]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t row = min(c0 + 8 * q + j, m - 1);                     // clamped: masked at use
            const float* yrow = Y + row * (int64_t)ld;
#pragma unroll
            for (int fb = 0; fb < NFB - 1; ++fb) fr[j][fb] = yrow[16 * fb + r];
            fr[j][NFB - 1] = yrow[last_col];
        }
    };
    if (row_lo < row_hi) load_chunk(row_lo, cur);
    for (int64_t c0 = row_lo; c0 < row_hi; c0 += stride) {
        if (c0 + stride < row_hi) load_chunk(c0 + stride, nxt);
        gram_bf16x8 hi[NFB], mid[NFB], lo[NFB];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float rmask = (c0 + 8 * q + j < row_hi) ? 1.f : 0.f;
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) {
                float x = cur[j][fb] * rmask;
                if (fb == NFB - 1) x *= col_mask_last;
                if (fb == 0 && bias && r == 0) x = rmask;                       // column 0 reads as 1 (wmf_model.py:331)
                const __bf16 h = (__bf16)x;
                const float r1 = x - (float)h;
                const __bf16 md = (__bf16)r1;
                hi[fb][j] = h; mid[fb][j] = md; lo[fb][j] = (__bf16)(r1 - (float)md);
            }
        }
        // product-major within a block row: consecutive MFMAs write different accumulators (six in a row on one tile wait
        // for each other's result)
        int t0 = 0;
#pragma unroll
        for (int bi = 0; bi < NFB; ++bi) {
#pragma unroll
            for (int bj = bi; bj < NFB; ++bj) acc[t0 + bj - bi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lo[bi], hi[bj], acc[t0 + bj - bi], 0, 0, 0);
#pragma unroll
            for (int bj = bi; bj < NFB; ++bj) acc[t0 + bj - bi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mid[bi], mid[bj], acc[t0 + bj - bi], 0, 0, 0);
#pragma unroll
            for (int bj = bi; bj < NFB; ++bj) acc[t0 + bj - bi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi[bi], lo[bj], acc[t0 + bj - bi], 0, 0, 0);
#pragma unroll
            for (int bj = bi; bj < NFB; ++bj) acc[t0 + bj - bi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mid[bi], hi[bj], acc[t0 + bj - bi], 0, 0, 0);
#pragma unroll
            for (int bj = bi; bj < NFB; ++bj) acc[t0 + bj - bi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi[bi], mid[bj], acc[t0 + bj - bi], 0, 0, 0);
#pragma unroll
            for (int bj = bi; bj < NFB; ++bj) acc[t0 + bj - bi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi[bi], hi[bj], acc[t0 + bj - bi], 0, 0, 0);
            t0 += NFB - bi;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) cur[j][fb] = nxt[j][fb];
    }
    float* out = partial + (int64_t)blockIdx.x * NT * 256;                     // partial layout: [wave][tile][reg][lane]
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) out[(t * 4 + reg) * 64 + lane] = acc[t][reg];
}

template <int NFB>
__global__ __launch_bounds__(64) void gram6_kernel(const float* __restrict__ Y, int64_t m, int f, int ld, int bias,
                                                   float* __restrict__ partial, int64_t steps_per_wave) {
    gram_body6<NFB>(Y, m, f, ld, bias, partial, blockIdx.x, gridDim.x);
}

// NSPLIT = 4 (f > 144: 136 tiles do not fit one wave's registers): the tile quarters are the FOUR WAVES OF ONE WORKGROUP, which
// walk the same rows at the same time -- one of them brings a row in from HBM, the other three find it in the CU's L1 / L2
// (round 2 launched the quarters as four separate workgroups: 8.26 GB of counted traffic for a 2.05 GB matrix at cfg5s).
template <int NFB, int NSPLIT>
__global__ __launch_bounds__(NSPLIT == 1 ? 64 : 256) void gram_kernel(const float* __restrict__ Y, int64_t m, int f, int ld, int bias,
                                                  float* __restrict__ partial, int64_t steps_per_wave) {
    const int64_t nsteps = (m + 3) / 4;
    int64_t lo = (int64_t)blockIdx.x * steps_per_wave;
    int64_t hi = lo + steps_per_wave;
    if (hi > nsteps) hi = nsteps;
    if (lo > hi) lo = hi;
    if constexpr (NSPLIT == 1) {
        gram_body<NFB, 1, 0>(Y, m, f, ld, bias, partial, lo, hi);
    } else {
        switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
            case 0: gram_body<NFB, NSPLIT, 0>(Y, m, f, ld, bias, partial, lo, hi); break;
            case 1: gram_body<NFB, NSPLIT, 1>(Y, m, f, ld, bias, partial, lo, hi); break;
            case 2: gram_body<NFB, NSPLIT, 2>(Y, m, f, ld, bias, partial, lo, hi); break;
            default: gram_body<NFB, NSPLIT, 3>(Y, m, f, ld, bias, partial, lo, hi); break;
        }
    }
}

// G_sum[i][j] (fp64) = sum over waves of the partial tile element; symmetric fill.  Two stages so
// that the sum over up to 1024 waves is spread over the chip: stage 1 sums a slice of the waves
// per (element, slice), stage 2 adds the slices in a fixed order (reproducible).
#define WMF_GRAM_SLICES 32
__global__ __launch_bounds__(256) void gram_reduce1_kernel(const float* __restrict__ partial, int nwaves, int f, int nfb,
                                                           double* __restrict__ slices) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= f * f) return;
    int i = e / f, j = e % f;
    if (i > j) { int tmp = i; i = j; j = tmp; }
    const int bi = i >> 4, bj = j >> 4, ri = i & 15, cj = j & 15;
    // index of tile (bi, bj), bi <= bj, in row-major enumeration of the upper triangle
    const int t = bi * nfb - (bi * (bi - 1)) / 2 + (bj - bi);
    const int lane = cj + 16 * (ri >> 2), reg = ri & 3;
    const int nt = nfb * (nfb + 1) / 2;
    const float* p = partial + ((int64_t)t * 4 + reg) * 64 + lane;
    const int per = (nwaves + WMF_GRAM_SLICES - 1) / WMF_GRAM_SLICES;
    const int w0 = blockIdx.y * per, w1 = min(nwaves, w0 + per);
    double s = 0.0;
    for (int w = w0; w < w1; ++w) s += (double)p[(int64_t)w * nt * 256];
    slices[(int64_t)blockIdx.y * f * f + e] = s;
}

__global__ __launch_bounds__(256) void gram_reduce2_kernel(const double* __restrict__ slices, int f, double* __restrict__ G) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= f * f) return;
    double s = 0.0;
#pragma unroll 8
    for (int k = 0; k < WMF_GRAM_SLICES; ++k) s += slices[(int64_t)k * f * f + e];
    G[e] = s;
}

template <int NFB>
static int launch_gram_nfb(const float* Y, int64_t m, int f, int ld, int bias, float* partial, int nwaves,
                           hipStream_t st) {
    const int64_t nsteps = (m + 3) / 4;
    const int64_t spw = (nsteps + nwaves - 1) / nwaves;
    if constexpr (NFB >= 7 && NFB <= 9) {                     // wide and still one accumulator set per wave: split-bf16 products
        if (!(wmf_debug_flags & 131072)) {                    // (debug flag 131072: the f32 MFMA kernel)
            static const char* nm = wmf_kname("gram6_kernel<%d>", NFB);
            WMF_LAUNCH(nm, (gram6_kernel<NFB>), dim3(nwaves), dim3(64), 0, st, Y, m, f, ld, bias, partial, spw);
            return 0;
        }
    }
    if constexpr (NFB <= 9) {
        static const char* nm = wmf_kname("gram_kernel<%d, 1>", NFB);
        WMF_LAUNCH(nm, (gram_kernel<NFB, 1>), dim3(nwaves), dim3(64), 0, st, Y, m, f, ld, bias, partial, spw);
    } else {
        static const char* nm = wmf_kname("gram_kernel<%d, 4>", NFB);
        WMF_LAUNCH(nm, (gram_kernel<NFB, 4>), dim3(nwaves), dim3(256), 0, st, Y, m, f, ld, bias, partial, spw);
    }
    return 0;
}

int wmf_gram_max_waves(int f) {            // keep the per-wave partial tiles within 64 MiB
    const int64_t nfb = (f + 15) / 16, nt = nfb * (nfb + 1) / 2;
    int64_t cap = ((int64_t)64 << 20) / (nt * 1024);
    if (cap > WMF_GRAM_MAX_WAVES) cap = WMF_GRAM_MAX_WAVES;
    return (int)(cap < 64 ? 64 : cap);
}

int wmf_gram_nwaves(int64_t m, int f) {
    int64_t nsteps = (m + 3) / 4;
    int64_t want = (nsteps + 31) / 32;     // at least 32 steps (128 rows) per wave
    if (want < 1) want = 1;
    const int cap = wmf_gram_max_waves(f);
    if (want > cap) want = cap;
    // every wave gets the same share of the rows up front, so the count is a whole number of waves per SIMD (256 CUs x 4):
    // 1456 waves of the wide kernel (one resident wave per SIMD) ran as two rounds, the second 42 % full
    if (want > 1024) want -= want % 1024;
    return (int)want;
}

int wmf_launch_gram(const float* Y, int64_t m, int f, int ld, int bias, double* G_sum, float* partial, double* slices,
                    hipStream_t st) {
    if (m <= 0) return hipMemsetAsync(G_sum, 0, (size_t)f * f * sizeof(double), st) == hipSuccess ? 0 : -1;
    const int nfb = (f + 15) / 16;
    const int nwaves = wmf_gram_nwaves(m, f);
    switch (nfb) {
#define C(N) case N: launch_gram_nfb<N>(Y, m, f, ld, bias, partial, nwaves, st); break;
        C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17)
#undef C
        default: return -1;
    }
    WMF_LAUNCH("gram_reduce1_kernel", gram_reduce1_kernel, dim3((f * f + 255) / 256, WMF_GRAM_SLICES), dim3(256), 0, st, partial,
               nwaves, f, nfb, slices);
    WMF_LAUNCH("gram_reduce2_kernel", gram_reduce2_kernel, dim3((f * f + 255) / 256), dim3(256), 0, st, slices, f, G_sum);
    return 0;
}

// ------------------------------------------------------------------------------------- factorize
// Single workgroup.  A (fp64, row stride lda odd) lives in LDS when it fits, else in the global
// workspace.  Right-looking Cholesky with one thread per trailing column, then the in-place
// inverse of the lower triangle (column by column from the last), then the two fp32 outputs.
// USE_LDS is a template parameter so that the matrix pointer keeps its address space: through a generic
// pointer every access becomes a flat_load that waits for both memory counters (measured 10x slower).
template <bool USE_LDS>
__global__ __launch_bounds__(256) void factorize_kernel(const double* __restrict__ G, int f, int ld, double lambda,
                                                        float* __restrict__ Wwhite, float* __restrict__ Wunwhite,
                                                        int32_t* __restrict__ info, double* __restrict__ gA) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int lda = f | 1;
    const int t = threadIdx.x;
    // the flag lives behind the matrix in the dynamic region (no static LDS in front of it)
    volatile int& s_fail = *reinterpret_cast<volatile int*>(smem_raw + (USE_LDS ? (size_t)f * lda * sizeof(double) : 0));
    auto body = [&](auto* A) {
    if (t == 0) s_fail = 0;
    for (int e = t; e < f * f; e += 256) {
        const int i = e / f, j = e % f;
        A[i * lda + j] = G[e] + (i == j ? lambda : 0.0);
    }
    __syncthreads();
    // ---- Cholesky (lower).  Column k: scale, then thread j owns trailing column j.
    for (int k = 0; k < f; ++k) {
        const double dk = A[k * lda + k];
        if (!(dk > 0.0)) {            // same value seen by every thread: uniform exit
            if (t == 0) s_fail = k + 1;
            break;
        }
        const double sk = sqrt(dk), inv = 1.0 / sk;
        __syncthreads();              // everyone has read A[k][k] before it is overwritten
        for (int i = k + 1 + t; i < f; i += 256) A[i * lda + k] *= inv;
        if (t == 0) A[k * lda + k] = sk;
        __syncthreads();
        {
            // trailing update A[i][j] -= L[i][k] L[j][k], i >= j > k.  P threads share a column
            // (rows i = j + part, j + part + P, ...); four rows per trip so the LDS reads overlap.
            const int ncol = f - k - 1;
            if (ncol > 0) {
                const int P = ncol >= 256 ? 1 : min(256 / ncol, 8);
                const int part = t / ncol;
                for (int j = k + 1 + (t % ncol); j < f && part < P; j += (ncol >= 256 ? 256 : f)) {
                    const double ljk = A[j * lda + k];
                    for (int i = j + part; i < f; i += 4 * P) {
                        const int i1 = i + P, i2 = i + 2 * P, i3 = i + 3 * P;
                        const double a0 = A[i * lda + k], c0 = A[i * lda + j];
                        const double a1 = i1 < f ? A[i1 * lda + k] : 0.0, c1 = i1 < f ? A[i1 * lda + j] : 0.0;
                        const double a2 = i2 < f ? A[i2 * lda + k] : 0.0, c2 = i2 < f ? A[i2 * lda + j] : 0.0;
                        const double a3 = i3 < f ? A[i3 * lda + k] : 0.0, c3 = i3 < f ? A[i3 * lda + j] : 0.0;
                        A[i * lda + j] = c0 - a0 * ljk;
                        if (i1 < f) A[i1 * lda + j] = c1 - a1 * ljk;
                        if (i2 < f) A[i2 * lda + j] = c2 - a2 * ljk;
                        if (i3 < f) A[i3 * lda + j] = c3 - a3 * ljk;
                    }
                }
            }
        }
        __syncthreads();
    }
    __syncthreads();
    const int fail = s_fail;
    if (t == 0 && fail) *info = fail;          // sticky: only the caller resets it (a later success must not hide a failure)
    if (fail) {                       // poison nothing: write zero transforms so downstream stays finite
        for (int e = t; e < f * ld; e += 256) { Wwhite[e] = 0.f; Wunwhite[e] = 0.f; }
        return;
    }
    // ---- in-place inverse of lower-triangular L (unblocked trti2, lower, non-unit):
    // for j = f-1 .. 0:  X[j][j] = 1/L[j][j];  X[i][j] = -X[j][j] * sum_{k=j+1..i} X[i][k] L[k][j]  (i > j)
    for (int j = f - 1; j >= 0; --j) {
        const double xjj = 1.0 / A[j * lda + j];
        double y0 = 0.0;
        const int i0 = j + 1 + t;      // f <= 260 and 256 threads: at most 2 rows per thread
        const int i1 = i0 + 256;
        double y1 = 0.0;
        if (i0 < f) for (int k = j + 1; k <= i0; ++k) y0 += A[i0 * lda + k] * A[k * lda + j];
        if (i1 < f) for (int k = j + 1; k <= i1; ++k) y1 += A[i1 * lda + k] * A[k * lda + j];
        __syncthreads();              // all reads of column j done before it is overwritten
        if (i0 < f) A[i0 * lda + j] = -xjj * y0;
        if (i1 < f) A[i1 * lda + j] = -xjj * y1;
        if (t == 0) A[j * lda + j] = xjj;
        __syncthreads();
    }
    // ---- outputs: Wunwhite[a][b] = Linv[a][b] (a >= b), Wwhite[a][b] = Linv[b][a] (b >= a)
    for (int e = t; e < f * ld; e += 256) {
        const int a = e / ld, b = e % ld;
        float wu = 0.f, ww = 0.f;
        if (b < f) {
            if (a >= b) wu = (float)A[a * lda + b];
            if (b >= a) ww = (float)A[b * lda + a];
        }
        Wunwhite[e] = wu;
        Wwhite[e] = ww;
    }
    };
    if constexpr (USE_LDS) body(reinterpret_cast<double*>(smem_raw));
    else body(gA);
}

// ---- f <= 64: one wave, one lane per matrix row, the whole fp64 matrix in registers ------------
// Right-looking Cholesky with the pivot column broadcast by v_readlane (two halves per double), then
// lane j builds column j of L^-1 by forward substitution.  No LDS, no barriers: ~10x faster than the
// workgroup version at f = 64, where 2 x 64 barrier-separated steps are pure latency.
__device__ __forceinline__ double rl_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
template <int K, int FP>
__device__ __forceinline__ void chol64_col(double (&a)[FP], bool& ok) {
    const double dk = rl_f64(a[K], K);
    if (!(dk > 0.0)) ok = false;
    const double inv = 1.0 / sqrt(dk);
    a[K] *= inv;                                    // lane K: sqrt(dk); lanes > K: L[i][K]
#pragma unroll
    for (int j = K + 1; j < FP; ++j) a[j] -= a[K] * rl_f64(a[K], j);
}
template <int FP, int... Ks>
__device__ __forceinline__ void chol64_sweep(double (&a)[FP], bool& ok, std::integer_sequence<int, Ks...>) {
    (chol64_col<Ks, FP>(a, ok), ...);
}
template <int I, int FP>
__device__ __forceinline__ void inv64_row(const double (&a)[FP], double (&x)[FP], int lane) {
    double s = (I == lane) ? 1.0 : 0.0;             // X[I][lane] = (delta - sum_{k<I} L[I][k] X[k][lane]) / L[I][I]
#pragma unroll
    for (int k = 0; k < I; ++k) s -= rl_f64(a[k], I) * x[k];
    x[I] = s / rl_f64(a[I], I);
}
template <int FP, int... Is>
__device__ __forceinline__ void inv64_sweep(const double (&a)[FP], double (&x)[FP], int lane, std::integer_sequence<int, Is...>) {
    (inv64_row<Is, FP>(a, x, lane), ...);
}

template <int FP>
__global__ __launch_bounds__(64, 1) void factorize64_kernel(const double* __restrict__ G, int f, int ld, double lambda,
                                                            float* __restrict__ Wwhite, float* __restrict__ Wunwhite,
                                                            int32_t* __restrict__ info) {
    const int lane = threadIdx.x;
    double a[FP];
#pragma unroll
    for (int j = 0; j < FP; ++j) {
        double v = (lane == j) ? 1.0 : 0.0;         // identity padding for rows / columns >= f
        if (lane < f && j < f) v = G[lane * f + j] + (lane == j ? lambda : 0.0);
        a[j] = v;
    }
    bool ok = true;
    chol64_sweep<FP>(a, ok, std::make_integer_sequence<int, FP>{});
    if (lane == 0 && !ok) *info = 1;           // sticky (see factorize_kernel)
    double x[FP];
    inv64_sweep<FP>(a, x, lane, std::make_integer_sequence<int, FP>{});
    // lane j holds column j of X = L^-1:  Wunwhite[i][j] = X[i][j],  Wwhite[j][i] = X[i][j]
    if (lane < f) {
#pragma unroll
        for (int i = 0; i < FP; ++i) {
            if (i < f) {
                const float v = ok ? (float)x[i] : 0.f;
                Wunwhite[i * ld + lane] = v;
                Wwhite[lane * ld + i] = v;
            }
        }
    }
    for (int e = lane; e < f * (ld - f); e += 64) {                      // zero padding columns [f, ld)
        const int row = e / (ld - f), col = f + e % (ld - f);
        Wunwhite[row * ld + col] = 0.f;
        Wwhite[row * ld + col] = 0.f;
    }
}

// ---- f <= 64, blocked: 16 x 16 blocks, fp64 MFMA for every block product -------------------------------
// The register-resident version above is correct but 32 000 instructions long (every (column, row) pair of both
// sweeps is unrolled): four times the instruction cache, so it runs at fetch speed (0.12 ms).  Here only the
// 16 x 16 diagonal blocks are factored and inverted with unrolled scalar code (one copy, called per block); panels,
// trailing updates and the block forward substitution of L^-1 are v_mfma_f64_16x16x4_f64 products on tiles in LDS.
//   A operand: lane (r = l & 15, q = l >> 4) supplies A[r][4 kk + q];  B: B[4 kk + q][r];
//   D: lane holds D[q + 4 v][r] in element v (the f64 form's row order differs from the f32 one).
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define WMF_MFMA16_F64(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)
constexpr int FZ_LD = 66;                              // row stride (doubles) of the 64 x 64 LDS images

// Tiles are 16 x 16 windows of row-major images with row strides la / lb / lt (doubles).
// D (+)= A . B^T : D[m][n] = sum_k A[m][k] B[n][k]
__device__ __forceinline__ f64x4 fz_mul_abt(const double* A, int la, const double* B, int lb, f64x4 c, int r, int q, double sign) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) c = WMF_MFMA16_F64(sign * A[r * la + 4 * kk + q], B[r * lb + 4 * kk + q], c);
    return c;
}
// D (+)= A . B : D[m][n] = sum_k A[m][k] B[k][n]
__device__ __forceinline__ f64x4 fz_mul_ab(const double* A, int la, const double* B, int lb, f64x4 c, int r, int q, double sign) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) c = WMF_MFMA16_F64(sign * A[r * la + 4 * kk + q], B[(4 * kk + q) * lb + r], c);
    return c;
}
__device__ __forceinline__ f64x4 fz_load(const double* T, int lt, int r, int q) {
    return f64x4{T[q * lt + r], T[(q + 4) * lt + r], T[(q + 8) * lt + r], T[(q + 12) * lt + r]};
}
__device__ __forceinline__ void fz_store(double* T, int lt, const f64x4& c, int r, int q) {
#pragma unroll
    for (int v = 0; v < 4; ++v) T[(q + 4 * v) * lt + r] = c[v];
}
// Cholesky of the 16 x 16 block D (lower part read) -> L into Lt (upper part zeroed), L^-1 into Xt.  Lane i < 16
// owns row i of D and L; lane j builds column j of the inverse.  One copy of the unrolled code for all blocks.
__device__ __noinline__ bool fz_diag(const double* D, double* Lt, double* Xt, int lt, int lane) {
    const int row = lane & 15;
    double a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = D[row * lt + j];
    bool ok = true;
    double rinv[16];                                               // 1 / L[K][K]: the substitution below multiplies by it
#pragma unroll
    for (int K = 0; K < 16; ++K) {
        const double dk = rl_f64(a[K], K);
        if (!(dk > 0.0)) ok = false;
        // 1 / sqrt(dk): hardware estimate + two Newton steps (full double precision; the library sqrt and divide are
        // each a long dependent sequence, and this chain is serial over the 16 columns)
        double y = __builtin_amdgcn_rsq(dk);
        y = y * (1.5 - 0.5 * dk * y * y);
        y = y * (1.5 - 0.5 * dk * y * y);
        rinv[K] = y;
        a[K] *= y;                                                  // lane K: sqrt(dk); lanes > K: L[i][K]
#pragma unroll
        for (int j = K + 1; j < 16; ++j) a[j] -= a[K] * rl_f64(a[K], j);
    }
    double x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        double sacc = (i == row) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; ++k) sacc -= rl_f64(a[k], i) * x[k];
        x[i] = sacc * rinv[i];
    }
    if (lane < 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            Lt[row * lt + j] = (j <= row) ? a[j] : 0.0;           // row `row` of L
            Xt[j * lt + row] = x[j];                               // column `row` of L^-1 (zero above the diagonal)
        }
    }
    return ok;
}

__global__ __launch_bounds__(64, 1) void factorize64m_kernel(const double* __restrict__ G, int f, int ld, double lambda,
                                                             float* __restrict__ Wwhite, float* __restrict__ Wunwhite,
                                                             int32_t* __restrict__ info) {
    __shared__ __attribute__((aligned(16))) double Gs[64 * FZ_LD];   // working matrix, becomes L (block lower triangle)
    __shared__ __attribute__((aligned(16))) double Xs[64 * FZ_LD];   // L^-1
    __shared__ __attribute__((aligned(16))) double Ts[16 * FZ_LD];   // one scratch tile
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    const int nb = (f + 15) >> 4;
#pragma unroll 16
    for (int i = 0; i < 64; ++i) {                                   // row i, column `lane`: 16 independent loads in flight
        const int j = lane;
        double v = (i == j) ? 1.0 : 0.0;                             // identity padding for rows / columns >= f
        const double gv = G[min(i, f - 1) * f + min(j, f - 1)];
        if (i < f && j < f) v = gv + (i == j ? lambda : 0.0);
        Gs[i * FZ_LD + j] = v;
        Xs[i * FZ_LD + j] = 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    bool ok = true;
    auto tile = [&](double* M, int bi, int bj) { return M + (16 * bi) * FZ_LD + 16 * bj; };
    auto sync = [&]() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    for (int p = 0; p < nb; ++p) {
        ok = fz_diag(tile(Gs, p, p), tile(Gs, p, p), tile(Xs, p, p), FZ_LD, lane) && ok;     // in place: row i is read before it is written
        sync();
        for (int i = p + 1; i < nb; ++i) {                           // L_ip = G_ip . (L_pp^-1)^T
            const f64x4 c = fz_mul_abt(tile(Gs, i, p), FZ_LD, tile(Xs, p, p), FZ_LD, f64x4{0.0, 0.0, 0.0, 0.0}, r, q, 1.0);
            sync();
            fz_store(tile(Gs, i, p), FZ_LD, c, r, q);
            sync();
        }
        for (int i = p + 1; i < nb; ++i)                             // G_ij -= L_ip . L_jp^T,  p < j <= i
            for (int j = p + 1; j <= i; ++j) {
                const f64x4 c = fz_mul_abt(tile(Gs, i, p), FZ_LD, tile(Gs, j, p), FZ_LD, fz_load(tile(Gs, i, j), FZ_LD, r, q), r, q, -1.0);
                fz_store(tile(Gs, i, j), FZ_LD, c, r, q);
            }
        sync();
    }
    // block forward substitution: X_ip = -X_ii . sum_{k = p .. i-1} L_ik X_kp
    for (int p = 0; p < nb; ++p)
        for (int i = p + 1; i < nb; ++i) {
            f64x4 sacc = f64x4{0.0, 0.0, 0.0, 0.0};
            for (int k = p; k < i; ++k) sacc = fz_mul_ab(tile(Gs, i, k), FZ_LD, tile(Xs, k, p), FZ_LD, sacc, r, q, 1.0);
            fz_store(Ts, FZ_LD, sacc, r, q);
            sync();
            const f64x4 c = fz_mul_ab(tile(Xs, i, i), FZ_LD, Ts, FZ_LD, f64x4{0.0, 0.0, 0.0, 0.0}, r, q, -1.0);
            fz_store(tile(Xs, i, p), FZ_LD, c, r, q);
            sync();
        }
    if (lane == 0 && !ok) *info = 1;           // sticky (see factorize_kernel)
    // Wunwhite[i][j] = X[i][j],  Wwhite[j][i] = X[i][j]; padding columns [f, ld) zero
#pragma unroll 8
    for (int i = 0; i < 64; ++i) {
        if (i >= f) break;
        for (int j = lane; j < ld; j += 64) {                        // ld <= 64 here: one pass
            float wu = 0.f, ww = 0.f;
            if (j < f && ok) { wu = (float)Xs[i * FZ_LD + j]; ww = (float)Xs[j * FZ_LD + i]; }
            Wunwhite[i * ld + j] = wu;
            Wwhite[i * ld + j] = ww;
        }
    }
}

// ---- 64 < f <= 272, blocked: the same algorithm with the two images in a global workspace (L2 resident) and the
// independent tile products of each step dealt to the eight waves of one workgroup.  The columns of L^-1 are
// independent of each other, so each wave substitutes its own columns without further barriers.
__global__ __launch_bounds__(512, 1) void factorize_blocked_kernel(const double* __restrict__ G, int f, int ld, double lambda,
                                                                   float* __restrict__ Wwhite, float* __restrict__ Wunwhite,
                                                                   int32_t* __restrict__ info, double* __restrict__ work) {
    __shared__ __attribute__((aligned(16))) double Ts[8][16 * 18];   // one scratch tile per wave
    __shared__ int bad;
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 15, q = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nb = (f + 15) >> 4, fp = 16 * nb, LD = fp + 2;
    double* A = work;                                                // fp x LD: working matrix, becomes L
    double* X = work + (size_t)fp * LD;                              // fp x LD: L^-1
    if (tid == 0) bad = 0;
    for (int e = tid; e < fp * fp; e += 512) {
        const int i = e / fp, j = e % fp;
        double v = (i == j) ? 1.0 : 0.0;                             // identity padding for rows / columns >= f
        if (i < f && j < f) v = G[i * f + j] + (i == j ? lambda : 0.0);
        A[i * LD + j] = v;
        X[i * LD + j] = 0.0;
    }
    __syncthreads();
    auto tile = [&](double* M, int bi, int bj) { return M + (size_t)(16 * bi) * LD + 16 * bj; };
    for (int p = 0; p < nb; ++p) {
        if (wave == 0) {
            const bool ok = fz_diag(tile(A, p, p), tile(A, p, p), tile(X, p, p), LD, lane);
            if (!ok && lane == 0) bad = 1;
        }
        __syncthreads();
        for (int i = p + 1 + wave; i < nb; i += 8) {                 // L_ip = A_ip . (L_pp^-1)^T, in place
            const f64x4 c = fz_mul_abt(tile(A, i, p), LD, tile(X, p, p), LD, f64x4{0.0, 0.0, 0.0, 0.0}, r, q, 1.0);
            fz_store(tile(A, i, p), LD, c, r, q);
        }
        __syncthreads();
        int t = 0;
        for (int i = p + 1; i < nb; ++i)                             // A_ij -= L_ip . L_jp^T,  p < j <= i
            for (int j = p + 1; j <= i; ++j, ++t) {
                if ((t & 7) != wave) continue;
                const f64x4 c = fz_mul_abt(tile(A, i, p), LD, tile(A, j, p), LD, fz_load(tile(A, i, j), LD, r, q), r, q, -1.0);
                fz_store(tile(A, i, j), LD, c, r, q);
            }
        __syncthreads();
    }
    // X_ip = -X_ii . sum_{k = p .. i-1} L_ik X_kp, column p by wave p mod 8
    double* ts = Ts[wave];
    for (int p = wave; p < nb; p += 8)
        for (int i = p + 1; i < nb; ++i) {
            f64x4 sacc = f64x4{0.0, 0.0, 0.0, 0.0};
            for (int k = p; k < i; ++k) sacc = fz_mul_ab(tile(A, i, k), LD, tile(X, k, p), LD, sacc, r, q, 1.0);
            fz_store(ts, 18, sacc, r, q);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const f64x4 c = fz_mul_ab(tile(X, i, i), LD, ts, 18, f64x4{0.0, 0.0, 0.0, 0.0}, r, q, -1.0);
            fz_store(tile(X, i, p), LD, c, r, q);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    __syncthreads();
    const bool ok = bad == 0;
    if (tid == 0 && !ok) *info = 1;            // sticky (see factorize_kernel)
    for (int e = tid; e < f * ld; e += 512) {
        const int i = e / ld, j = e % ld;
        float wu = 0.f, ww = 0.f;
        if (j < f && ok) { wu = (float)X[i * LD + j]; ww = (float)X[j * LD + i]; }
        Wunwhite[e] = wu;
        Wwhite[e] = ww;
    }
}

int wmf_launch_factorize(const double* G_sum, int f, int ld, double lambda, float* Wwhite, float* Wunwhite,
                         int32_t* info, double* gA, hipStream_t st) {
    if (f <= 64 && !(wmf_debug_flags & 512)) {       // blocked single-wave version (fp64 MFMA)
        WmfProfScope ps("factorize64m_kernel", st);
        hipLaunchKernelGGL(factorize64m_kernel, dim3(1), dim3(64), 0, st, G_sum, f, ld, lambda, Wwhite, Wunwhite, info);
        return 0;
    }
    if (f <= 64) {                                   // register-resident single-wave version (debug flag 512: A/B timing)
        WmfProfScope ps("factorize64_kernel", st);
        if (f <= 16) hipLaunchKernelGGL(factorize64_kernel<16>, dim3(1), dim3(64), 0, st, G_sum, f, ld, lambda, Wwhite, Wunwhite, info);
        else if (f <= 32) hipLaunchKernelGGL(factorize64_kernel<32>, dim3(1), dim3(64), 0, st, G_sum, f, ld, lambda, Wwhite, Wunwhite, info);
        else if (f <= 48) hipLaunchKernelGGL(factorize64_kernel<48>, dim3(1), dim3(64), 0, st, G_sum, f, ld, lambda, Wwhite, Wunwhite, info);
        else hipLaunchKernelGGL(factorize64_kernel<64>, dim3(1), dim3(64), 0, st, G_sum, f, ld, lambda, Wwhite, Wunwhite, info);
        return 0;
    }
    if (!(wmf_debug_flags & 512)) {                  // blocked workgroup version (fp64 MFMA); flag 512: the older kernel below
        WmfProfScope ps("factorize_blocked_kernel", st);
        hipLaunchKernelGGL(factorize_blocked_kernel, dim3(1), dim3(512), 0, st, G_sum, f, ld, lambda, Wwhite, Wunwhite, info, gA);
        return 0;
    }
    const int lda = f | 1;
    const size_t bytes = (size_t)f * lda * sizeof(double);
    const bool use_lds = bytes <= 150 * 1024;
    static bool attr_set = false;
    if (use_lds && !attr_set) {
        (void)hipFuncSetAttribute((const void*)factorize_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  150 * 1024 + 16);
        attr_set = true;
    }
    WmfProfScope ps("factorize_kernel", st);
    if (use_lds)
        hipLaunchKernelGGL(factorize_kernel<true>, dim3(1), dim3(256), bytes + 16, st, G_sum, f, ld, lambda, Wwhite, Wunwhite,
                           info, gA);
    else
        hipLaunchKernelGGL(factorize_kernel<false>, dim3(1), dim3(256), 16, st, G_sum, f, ld, lambda, Wwhite, Wunwhite, info, gA);
    return 0;
}

// ------------------------------------------------------------------------------------- transform
// out[row][:] = in~[row][:] . W.  A wave owns 16-row blocks.  Lane (r = l & 15, q = l >> 4) loads
// 16-byte pieces in[row0 + r][16 t + 4 q .. +3]; its element e is the A operand of MFMA step
// (t, e), whose k slot q then means k = 16 t + 4 q + e; the B operand is W[k][16 nb + r] read from
// the LDS copy of W (row stride ldw = 4 mod 8 dwords keeps the two k rows of a 32-lane half on
// different banks).  The sum over k is order independent, so this k permutation is free.
// NB0 / NBW: this launch produces the NBW output blocks starting at block NB0 (all of them when the whole W fits
// LDS; two or three column slices, each with its own launch, when it does not: f > 176).
template <int NFB, bool W_IN_LDS, int NB0 = 0, int NBW = NFB>
__global__ __launch_bounds__(512) void transform_kernel(const float* __restrict__ in, int64_t m, int f, int ld,
                                                        const float* __restrict__ W, int set_col0_one,
                                                        float* __restrict__ out, float* __restrict__ col0_out,
                                                        int64_t nblocks16) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Wl = reinterpret_cast<float*>(smem_raw);
    constexpr int KP = 16 * NFB;          // padded K (rows of W)
    constexpr int NP = 16 * NBW;          // padded N of this slice (columns 16 NB0 .. of W)
    constexpr int LDW = NP + 4;           // 4 mod 8 since NP is a multiple of 16
    const int tid = threadIdx.x;
    // nz[kb * NFB + nb] != 0 iff the 16 x 16 tile (kb, nb) of W has a non-zero entry.  Both matrices this kernel is
    // used with are triangular (L^-T, L^-1), so 36 of 81 tile products at f = 129 are skipped -- found from the data,
    // any W stays correct.
    __shared__ int nz[NFB * NBW];
    for (int e = tid; e < NFB * NBW; e += 512) nz[e] = 0;
    __syncthreads();
    for (int e = tid; e < KP * LDW; e += 512) {
        const int a = e / LDW, b = e % LDW, col = 16 * NB0 + b;
        const float v = (a < f && b < NP && col < f) ? W[a * ld + col] : 0.f;
        if constexpr (W_IN_LDS) Wl[e] = v;
        if (v != 0.f) nz[(a >> 4) * NBW + (b >> 4)] = 1;              // benign race: every writer stores 1
    }
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int nch = ld >> 2;
    // per k block t: which output blocks have a non-zero tile of W (wave-uniform bit mask, kept in SGPRs)
    unsigned tmask[NFB];
#pragma unroll
    for (int t = 0; t < NFB; ++t) {
        unsigned mk = 0;
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) mk |= nz[t * NBW + nb] ? (1u << nb) : 0u;
        tmask[t] = __builtin_amdgcn_readfirstlane(mk);
    }
    // A whole 16-row block (all NFB pieces of a lane's row) is requested at once, and the NEXT block of this wave is
    // requested before the MFMAs of the current one start: 2 x NFB 16-byte loads in flight per lane.  (One piece ahead,
    // as before, left a 10 M x 132 transform at 2.8 TB/s: two 1 KB wave loads in flight per wave hide no HBM latency.)
    // Beyond NFB = 11 (the column-sliced launches for f > 176) two blocks of pieces no longer fit the 256 registers of a
    // wave at two waves per SIMD: one block at a time there, all its pieces requested together.
    constexpr bool DBL = NFB <= 11;
    const int64_t stride = (int64_t)gridDim.x * 8;
    auto request = [&](int64_t blk, float4 (&x)[NFB]) {
        const int64_t row = blk * 16 + r;
        const float4* irow = reinterpret_cast<const float4*>(in + (row < m ? row : 0) * (int64_t)ld);   // loads are unconditional
#pragma unroll
        for (int t = 0; t < NFB; ++t) x[t] = irow[min(4 * t + q, nch - 1)];
    };
    auto block = [&](int64_t blk, float4 (&xc)[NFB], float4 (&xn)[NFB]) {
        if constexpr (DBL) { if (blk + stride < nblocks16) request(blk + stride, xn); }
        const int64_t row = blk * 16 + r;
        const bool rok = row < m;
        f32x4 acc[NBW];
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NFB; ++t) {
            const int c = 4 * t + q;                 // 16-byte piece index within the row
            const int k0 = 4 * c;
            // Rows past m compute garbage that is never stored (row r of A only reaches output row r), and every piece but
            // the last lies inside the f features, so only the last piece is masked (its load was clamped).
            float xe[4] = {xc[t].x, xc[t].y, xc[t].z, xc[t].w};
            if (t == NFB - 1) {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (!(c < nch && k0 + e < f)) xe[e] = 0.f;
            }
            if (t == 0 && set_col0_one && q == 0) {
                // (set_col0_one == 2, the split layout: col0_out holds {last feature, bias} pairs)
                if (rok && col0_out && NB0 == 0) col0_out[set_col0_one == 2 ? 2 * row + 1 : row] = xe[0];
                xe[0] = 1.f;
            }
            unsigned mv = tmask[t];
            asm volatile("" : "+v"(mv));                     // test the bits here: hoisted, the 81 branch conditions spill
            const unsigned mk = __builtin_amdgcn_readfirstlane(mv);
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                if (!(mk & (1u << nb))) continue;            // wave-uniform: an all-zero tile of W
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = k0 + e;
                    float b;
                    if constexpr (W_IN_LDS) {
                        b = Wl[k * LDW + 16 * nb + r];
                    } else {
                        const int col = 16 * (NB0 + nb) + r;
                        b = (k < f && col < f) ? W[k * ld + col] : 0.f;
                    }
                    acc[nb] = WMF_MFMA16(xe[e], b, acc[nb]);
                }
            }
        }
        // acc[nb][reg] = out[blk*16 + 4q + reg][16 nb + r]
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t orow = blk * 16 + 4 * q + reg;
            if (orow < m) {
                // split layout (set_col0_one == 2): packed body rows of f - 1 floats, feature f - 1 goes to the pairs
                const bool sp = set_col0_one == 2;
                const int wid = sp ? f - 1 : ld;
                float* o = out + orow * (int64_t)wid;
#pragma unroll
                for (int nb = 0; nb < NBW; ++nb) {
                    const int col = 16 * (NB0 + nb) + r;
                    if (col < wid) o[col] = acc[nb][reg];
                    else if (sp && col == f - 1) col0_out[2 * orow] = acc[nb][reg];
                }
            }
        }
    };
    int64_t blk = (int64_t)blockIdx.x * 8 + wv;
    if constexpr (DBL) {
        float4 xa[NFB], xb[NFB];
        if (blk < nblocks16) request(blk, xa);
        while (blk < nblocks16) {
            block(blk, xa, xb);
            blk += stride;
            if (blk >= nblocks16) break;
            block(blk, xb, xa);
            blk += stride;
        }
    } else {
        float4 xa[NFB];
        for (; blk < nblocks16; blk += stride) {
            request(blk, xa);
            block(blk, xa, xa);
        }
    }
}

// The same product with split-bf16 MFMAs for wide factors (f = 97 .. 144), where the f32 MFMAs bound the kernel above (180 of
// them per 16-row block at f = 129 against a 1.3 ms HBM floor for 10 M rows).  W is split ONCE into three bf16 planes, stored
// transposed in LDS (plane p, output column n, k contiguous: the eight k of a lane's 16x16x32 B operand are one ds_read_b128;
// row stride 176 bf16 = 88 dwords: the 16 lanes that share an LDS cycle of a ds_read_b128 -- {0-3, 12-15, 20-27} .. -- then hit
// 64 distinct banks; at 84 dwords three pairs of them collided and the kernel ran at half speed); the rows of `in` are split on
// the fly (a lane's A operand of K chunk c is the two 16-byte pieces 8 c + 2 q, 8 c + 2 q + 1 of its row).  Six MFMAs per
// (32-wide K chunk, output block) whose W tile is not all zero: 25 of 45 pairs for a triangular W at f = 129.
template <int NFB>
__global__ __launch_bounds__(512) void transform6_kernel(const float* __restrict__ in, int64_t m, int f, int ld,
                                                         const float* __restrict__ W, int set_col0_one,
                                                         float* __restrict__ out, float* __restrict__ col0_out,
                                                         int64_t nblocks16) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int NP = 16 * NFB;                 // output columns, padded
    constexpr int NKC = (NP + 31) / 32;          // K chunks of 32
    constexpr int KS = 176;                      // bf16 elements per LDS row (>= 32 NKC = 160): 88 dwords, see above
    static_assert(32 * NKC <= KS, "K does not fit the LDS row");
    __bf16* Wt = reinterpret_cast<__bf16*>(smem_raw);             // [3][NP][KS]
    __shared__ int nz[NKC * NFB];
    const int tid = threadIdx.x;
    for (int e = tid; e < NKC * NFB; e += 512) nz[e] = 0;
    __syncthreads();
    // set_col0_one == 3 / 4: the ROLLED whitened coordinates (include/wmf_hip.h) -- whitened feature c lives at position c - 1 and
    // feature 0 at position f - 1: column n of the matrix in LDS is column n + 1 of W when whitening (3), its row k is row k + 1
    // of W when the input is in those coordinates (4, the un-whitening); both mod f
    const bool roll_n = set_col0_one == 3, roll_k = set_col0_one == 4;
    for (int e = tid; e < NP * KS; e += 512) {
        const int n = e / KS, k = e % KS;
        const int ns = (roll_n && n < f) ? (n + 1 == f ? 0 : n + 1) : n, ks = (roll_k && k < f) ? (k + 1 == f ? 0 : k + 1) : k;
        const float v = (k < f && n < f) ? W[ks * ld + ns] : 0.f;
        const __bf16 h = (__bf16)v;
        const float r1 = v - (float)h;
        const __bf16 md = (__bf16)r1;
        Wt[e] = h;
        Wt[NP * KS + e] = md;
        Wt[2 * NP * KS + e] = (__bf16)(r1 - (float)md);
        if (v != 0.f) nz[(k >> 5) * NFB + (n >> 4)] = 1;          // benign race: every writer stores 1
    }
    __syncthreads();
    const int lane = tid & 63, wv = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int nch = ld >> 2;
    unsigned cmask[NKC];                         // per K chunk: output blocks with a non-zero tile of W
#pragma unroll
    for (int c = 0; c < NKC; ++c) {
        unsigned mk = 0;
#pragma unroll
        for (int nb = 0; nb < NFB; ++nb) mk |= nz[c * NFB + nb] ? (1u << nb) : 0u;
        cmask[c] = __builtin_amdgcn_readfirstlane(mk);
    }
    typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
    const bf16x8_t* Wl = reinterpret_cast<const bf16x8_t*>(Wt);    // 16-byte units: element (p, n, k8) at ((p NP + n) KS + 8 k8) / 8
    const int64_t stride = (int64_t)gridDim.x * 8;
    auto request = [&](int64_t blk, float4 (&x)[2 * NKC]) {
        const int64_t row = blk * 16 + r;
        const float4* irow = reinterpret_cast<const float4*>(in + (row < m ? row : 0) * (int64_t)ld);   // loads are unconditional
#pragma unroll
        for (int c = 0; c < NKC; ++c) {
            x[2 * c] = irow[min(8 * c + 2 * q, nch - 1)];
            x[2 * c + 1] = irow[min(8 * c + 2 * q + 1, nch - 1)];
        }
    };
    const bool in_one = set_col0_one >= 1 && set_col0_one <= 3;      // column 0 of the input reads as 1 (whitening of a bias model)
    const bool sp = set_col0_one == 2 || set_col0_one == 3;          // split layout: packed body rows of f - 1 floats + the pairs
    auto block = [&](int64_t blk, float4 (&xc)[2 * NKC]) {
        const int64_t row = blk * 16 + r;
        const bool rok = row < m;
        float bias_of_row = 0.f;                                     // (lanes q = 0: the bias of row blk 16 + r)
        f32x4 acc[NFB];
#pragma unroll
        for (int nb = 0; nb < NFB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NKC; ++c) {
            const int k0 = 32 * c + 8 * q;                       // first k of this lane's eight
            float xe[8] = {xc[2 * c].x, xc[2 * c].y, xc[2 * c].z, xc[2 * c].w, xc[2 * c + 1].x, xc[2 * c + 1].y, xc[2 * c + 1].z, xc[2 * c + 1].w};
            if (32 * c + 32 > f) {                               // only the last chunk(s) can run past the f features (loads were clamped)
#pragma unroll
                for (int e = 0; e < 8; ++e) if (!(k0 + e < f)) xe[e] = 0.f;
            }
            if (c == 0 && in_one && q == 0) {
                if (rok && col0_out) col0_out[sp ? 2 * row + 1 : row] = xe[0];                     // (split layout: the pairs)
                bias_of_row = xe[0];
                xe[0] = 1.f;
            }
            bf16x8_t ah, am, al;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const __bf16 h = (__bf16)xe[e];
                const float r1 = xe[e] - (float)h;
                const __bf16 md = (__bf16)r1;
                ah[e] = h; am[e] = md; al[e] = (__bf16)(r1 - (float)md);
            }
            unsigned mv = cmask[c];
            asm volatile("" : "+v"(mv));                         // test the bits here (hoisted, the branch conditions spill)
            const unsigned mk = __builtin_amdgcn_readfirstlane(mv);
#pragma unroll
            for (int nb = 0; nb < NFB; ++nb) {
                if (!(mk & (1u << nb))) continue;                // wave-uniform: an all-zero 32 x 16 tile of W
                const int at = ((16 * nb + r) * KS + 32 * c + 8 * q) / 8;
                const bf16x8_t bh = Wl[at], bm = Wl[NP * KS / 8 + at], bl = Wl[2 * NP * KS / 8 + at];
                f32x4 cc = acc[nb];
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, cc, 0, 0, 0);
                acc[nb] = cc;
            }
        }
        // acc[nb][reg] = out[blk*16 + 4q + reg][16 nb + r]
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int64_t orow = blk * 16 + 4 * q + reg;
            // (3) the bias of the output row, for its bits: it sits in lane 4 q + reg (q = 0 there)
            const unsigned bbits = set_col0_one == 3 ? (unsigned)__builtin_amdgcn_ds_bpermute((4 * q + reg) * 4, __builtin_bit_cast(int, bias_of_row)) : 0u;
            if (orow < m) {
                // split layout: packed body rows of f - 1 floats, feature f - 1 goes to the pairs
                const int wid = sp ? f - 1 : ld;
                float* o = out + orow * (int64_t)wid;
#pragma unroll
                for (int nb = 0; nb < NFB; ++nb) {
                    const int col = 16 * nb + r;
                    float v = acc[nb][reg];
                    // (3) bit 2 (col / 8) + col % 8 of the row's bias replaces the last mantissa bit of body positions 8 j, 8 j + 1
                    // (j < 16): the row kernels that hold 8 consecutive features per lane rebuild the bias from the row they
                    // gathered instead of fetching the pair (csrc/wmf_iter.hip); the value moves by at most one ulp
                    if (set_col0_one == 3 && col < 128 && (col & 7) < 2)
                        v = __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, v) & ~1u) | ((bbits >> (2 * (col >> 3) + (col & 7))) & 1u));
                    if (col < wid) o[col] = v;
                    else if (sp && col == f - 1) col0_out[2 * orow] = v;
                }
            }
        }
    };
    // one block of pieces at a time (two, as in transform_kernel, spill at 256 registers: the bf16 parts need room).  Two
    // 16-row blocks per trip sharing every B operand read (twelve MFMAs per three ds_read_b128 instead of six) were measured in
    // round 2: 256 registers + 244 bytes of scratch at NFB = 9, 3.9 ms per cfg3 half step against 2.4.  Requesting the next
    // block's pieces into this block's registers chunk by chunk, as soon as each is split (no spill, a block of arithmetic
    // between request and use): 2.25 against 2.06 ms for 10 M rows in the same process -- the two waves of a SIMD already
    // cover each other's wait, and the kernel moves 5.1 TB/s.
    float4 xa[2 * NKC];
    for (int64_t blk = (int64_t)blockIdx.x * 8 + wv; blk < nblocks16; blk += stride) {
        request(blk, xa);
        block(blk, xa);
    }
}

template <int NFB>
static void launch_transform6(const float* in, int64_t m, int f, int ld, const float* W, int set_col0_one, float* out,
                              float* col0_out, int64_t grid, int64_t nblk, hipStream_t st) {
    constexpr size_t lds = (size_t)3 * 16 * NFB * 176 * 2;
    static_assert(lds <= 158 * 1024, "W planes do not fit LDS");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)transform6_kernel<NFB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    static const char* nm = wmf_kname("transform6_kernel<%d>", NFB);
    WMF_LAUNCH(nm, (transform6_kernel<NFB>), dim3((unsigned)grid), dim3(512), lds, st, in, m, f, ld, W, set_col0_one, out,
               col0_out, nblk);
}

template <int NFB, int NB0, int NBW>
static void launch_transform_slice(const float* in, int64_t m, int f, int ld, const float* W, int set_col0_one, float* out,
                                   float* col0_out, int64_t grid, int64_t nblk, hipStream_t st) {
    constexpr size_t lds = (size_t)16 * NFB * (16 * NBW + 4) * 4;
    static_assert(lds <= 150 * 1024, "slice too wide");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)transform_kernel<NFB, true, NB0, NBW>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr_set = true;
    }
    static const char* nm = wmf_kname("transform_kernel<%d, true, %d, %d>", NFB, NB0, NBW);
    WMF_LAUNCH(nm, (transform_kernel<NFB, true, NB0, NBW>), dim3((unsigned)grid), dim3(512), lds, st, in, m, f, ld, W,
               set_col0_one, out, col0_out, nblk);
}

template <int NFB>
static void launch_transform_nfb(const float* in, int64_t m, int f, int ld, const float* W, int set_col0_one, float* out,
                                 float* col0_out, hipStream_t st) {
    constexpr int KP = 16 * NFB, LDW = KP + 4;
    constexpr size_t lds = (size_t)KP * LDW * 4;
    const int64_t nblk = (m + 15) / 16;
    int64_t grid = (nblk + 7) / 8;
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    if constexpr (NFB >= 7 && NFB <= 9) {                     // wide factors: split-bf16 products (debug flag 262144: f32 MFMAs)
        if (!(wmf_debug_flags & 262144)) {
            launch_transform6<NFB>(in, m, f, ld, W, set_col0_one, out, col0_out, grid, nblk, st);
            return;
        }
    }
    if constexpr (lds <= 150 * 1024) {
        launch_transform_slice<NFB, 0, NFB>(in, m, f, ld, W, set_col0_one, out, col0_out, grid, nblk, st);
    } else if (in != out) {
        // W does not fit LDS (f > 176): two or three column slices of W, one launch each; every launch reads the
        // input again (it may not alias the output) and writes its own output columns
        constexpr int H = (NFB + 1) / 2;
        if constexpr (H <= 8) {
            launch_transform_slice<NFB, 0, H>(in, m, f, ld, W, set_col0_one, out, col0_out, grid, nblk, st);
            launch_transform_slice<NFB, H, NFB - H>(in, m, f, ld, W, set_col0_one, out, col0_out, grid, nblk, st);
        } else {
            launch_transform_slice<NFB, 0, 6>(in, m, f, ld, W, set_col0_one, out, col0_out, grid, nblk, st);
            launch_transform_slice<NFB, 6, 6>(in, m, f, ld, W, set_col0_one, out, col0_out, grid, nblk, st);
            launch_transform_slice<NFB, 12, NFB - 12>(in, m, f, ld, W, set_col0_one, out, col0_out, grid, nblk, st);
        }
    } else {
        static const char* nm = wmf_kname("transform_kernel<%d, false, 0, %d>", NFB, NFB);
        WMF_LAUNCH(nm, (transform_kernel<NFB, false>), dim3((unsigned)grid), dim3(512), 0, st, in, m, f, ld, W,
                   set_col0_one, out, col0_out, nblk);
    }
}

int wmf_launch_transform(const float* in, int64_t m, int f, int ld, const float* W, int set_col0_one, float* out,
                         float* col0_out, hipStream_t st) {
    if (m <= 0) return 0;
    const int nfb = (f + 15) / 16;
    // set_col0_one == 2 inside the kernels: the split layout (wmf_internal.h) -- `out` is the packed body, col0_out the pairs;
    // 3 / 4 (callers' values, wmf_rolled_layout_supported): the rolled coordinates, transform6_kernel only
    if (set_col0_one == 3 || set_col0_one == 4) {
        if (!wmf_rolled_layout(f, ld)) return -1;
    } else {
        set_col0_one = set_col0_one ? (wmf_split_layout(f, ld) ? 2 : 1) : 0;
    }
    if ((set_col0_one == 2 || set_col0_one == 3) && (!col0_out || in == out)) return -3;
    switch (nfb) {
#define C(N) case N: launch_transform_nfb<N>(in, m, f, ld, W, set_col0_one, out, col0_out, st); break;
        C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17)
#undef C
        default: return -1;
    }
    return 0;
}
