// One half step in float64 (gfx950): the arithmetic of the reference's Pool variants, RecModel/wmf_model.py:242-309
// (recompute_factors_par / recompute_factors_bias_par and their *_intern row functions).  With a float64 (or integer) count
// matrix those keep every row result in float64 -- np.stack of the per-row np.linalg.solve outputs, no cast back to the model
// dtype -- so from the second half step on the reference's cores > 1 training runs on float64 factors, float64 Gramians and
// float64 row systems.  cores = 4 is the reference's DEFAULT (wmf_model.py:49-51) and SciPy matrices default to float64, so
// this is the path a drop-in caller lands on.  Round 3: register-blocked kernels instead of one read-modify-write of the
// system per stored entry.
//   gram64v2_kernel + gram64v2_reduce_kernel   G = Y~^T Y~ + lambda I      (:244 / :258; Y~ = Y with column 0 read as 1, :257)
//       a workgroup stages 16 rows of Y at a time in LDS (coalesced) and every thread keeps 4 x 4 blocks of the upper
//       triangle in registers (8 LDS reads of 16 bytes per 16 FMAs); block partial sums are added in a fixed order.
//   solve64v2_kernel<NB>   per row u: A = G + Y_u^T diag(w) Y_u,  b = Y_u^T (w + 1)   (:285-287 / :305-309), w = c_u - bias[idx]
//       for bias models (:279).  One 256-thread workgroup per row; the AUGMENTED upper triangle [A | b] lives in registers
//       as 4 x 4 blocks (NB per thread: 1 up to f = 84, 3 up to 148, 9 up to 260) from the first gathered entry to the
//       solution: entries are gathered 16 at a time into LDS and accumulated like the Gramian, then a right-looking blocked
//       Cholesky A = R^T R runs on the same registers -- per block column: the diagonal block's owner factors and inverts its
//       4 x 4 block, the owners of that block row form their R blocks (and publish them in a double-buffered LDS panel), the
//       rest subtract their rank-4 update -- with b riding along as one more block column (so R^-T b is there when the
//       factorisation ends), and the back substitution R x = y walks the block columns again with each R block still in the
//       registers of its owner.  Two barriers per block column and phase, no workspace.
//       np.linalg.solve is LU with partial pivoting; for a positive definite system Cholesky gives the same solution to a
//       few ulp of float64 (the tests hold both to 1e-10 of the float64 reference arithmetic).
//   solve64_lu_kernel      rows whose system is NOT positive definite (bias-adjusted weights below zero, :279) or not finite:
//       the factorisation above meets a non-positive pivot and hands the row over -- LU with partial pivoting (first maximum as
//       idamax, i.e. gesv), the right-hand side carried along, the row system in an L2-resident workspace slice; an exactly
//       singular system is counted and NaN-filled (the reference raises).  Rows without stored entries are zero (:274-276 /
//       :296-298).
#include "../../include/wmf_hip.h"
#include "wmf_internal.h"

// ---- 4 x 4 blocks of the (augmented) upper triangle, dealt to the 256 threads of a workgroup -------------------------------
// f4 = ceil(f / 4) block rows; row bi holds the blocks (bi, bi) .. (bi, f4 - 1) and, when RHS, one more at column f4: the
// right-hand side (its first column; the other three stay zero).  Block number b = n * 256 + thread, n < NB.
#define F64_R 16                                   /* gathered entries (or rows of Y) staged per pass */
#define F64_LR_D 32                                /* rows with at most this many entries take the low-rank form (solve64lr_kernel) */
__device__ __forceinline__ void f64_decode(int b, int f4, bool rhs, int& bi, int& bj) {
    int r = 0, rem = b;
    const int extra = rhs ? 1 : 0;
    while (r < f4 && rem >= f4 - r + extra) { rem -= f4 - r + extra; ++r; }
    bi = r; bj = r + rem;
}
static int f64_blocks(int f, bool rhs) { const int f4 = (f + 3) / 4; return f4 * (f4 + 1) / 2 + (rhs ? f4 : 0); }
static int f64_nb(int nblk) { return nblk <= 256 ? 1 : nblk <= 512 ? 2 : nblk <= 768 ? 3 : nblk <= 1280 ? 5 : 9; }

// acc[n] += sum_e (w_e a_e) b_e^T over the staged entries: a_e = ys[e][4 bi ..], b_e = ys[e][4 bj ..]; the right-hand-side
// block (bj == f4) takes b_e = (p_e, 0, 0, 0) un-weighted (column FP of the staged row holds p_e)
template <int NB>
__device__ __forceinline__ void f64_accumulate(double (&acc)[NB][16], const int (&bi)[NB], const int (&bj)[NB], const bool (&on)[NB],
                                               const double* __restrict__ ys, int ldy, const double* __restrict__ wv, int nvalid, int f4) {
    for (int e = 0; e < nvalid; ++e) {
        const double w = wv[e];
        const double* row = ys + e * ldy;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            if (!on[n]) continue;
            const double2 a01 = *reinterpret_cast<const double2*>(row + 4 * bi[n]), a23 = *reinterpret_cast<const double2*>(row + 4 * bi[n] + 2);
            const double2 b01 = *reinterpret_cast<const double2*>(row + 4 * bj[n]), b23 = *reinterpret_cast<const double2*>(row + 4 * bj[n] + 2);
            const double ws = bj[n] == f4 ? 1.0 : w;
            const double a[4] = {a01.x * ws, a01.y * ws, a23.x * ws, a23.y * ws};
            const double b[4] = {b01.x, b01.y, b23.x, b23.y};
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[n][4 * x + y] = __builtin_fma(a[x], b[y], acc[n][4 * x + y]);
        }
    }
}

// ---- Gramian: partial[wg][block][16] = upper-triangle blocks of Y~^T Y~ over this workgroup's rows -----------------------------
template <int NB>
__global__ __launch_bounds__(256) void gram64v2_kernel(const double* __restrict__ Y, int64_t m, int f, int bias,
                                                       double* __restrict__ partial, int64_t rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) double sm64[];
    const int t = threadIdx.x;
    const int f4 = (f + 3) >> 2, FP = 4 * f4, nblk = f4 * (f4 + 1) / 2;
    double* ys = sm64;                              // [F64_R][FP]
    double* wv = ys + F64_R * FP;                   // [F64_R] ones
    int bi[NB], bj[NB];
    bool on[NB];
    double acc[NB][16];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int b = n * 256 + t;
        on[n] = b < nblk;
        f64_decode(on[n] ? b : 0, f4, false, bi[n], bj[n]);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.0;
    }
    if (t < F64_R) wv[t] = 1.0;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(m, r0 + rows_per_block);
    for (int64_t c0 = r0; c0 < r1; c0 += F64_R) {
        const int nvalid = (int)min((int64_t)F64_R, r1 - c0);
        __syncthreads();                            // the previous pass is done with ys
        for (int i = t; i < nvalid * FP; i += 256) {
            const int e = i / FP, c = i - e * FP;
            ys[i] = c < f ? ((bias && c == 0) ? 1.0 : Y[(c0 + e) * f + c]) : 0.0;
        }
        __syncthreads();
        f64_accumulate<NB>(acc, bi, bj, on, ys, FP, wv, nvalid, f4 + 1);       // (f4 + 1: no right-hand-side block here)
    }
    double* out = partial + (int64_t)blockIdx.x * nblk * 16;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        if (!on[n]) continue;
        double2* o = reinterpret_cast<double2*>(out + (int64_t)(n * 256 + t) * 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = make_double2(acc[n][2 * i], acc[n][2 * i + 1]);
    }
}

// The Gramian on the matrix cores (f <= 144): eight waves per workgroup, each with up to PW of the 16 x 16 upper tiles (bi <= bj)
// of Y~^T Y~; a tile's two operands of v_mfma_f64_16x16x4_f64 are the same kind of fragment -- lane (r, q) holds
// Y~[row0 + q][16 b + r], for b = bi and b = bj -- read from an LDS image of sixteen rows (row stride 16 mod 32 doubles: the four
// rows of a read on disjoint banks; fragments straight from global memory made the texture path the bound: 96 eight-byte loads
// per four rows and workgroup, 13.5 ms at cfg3's size), the next sixteen rows fetched while the products run, one barrier per
// sixteen rows.  Results leave in the 4 x 4 block layout gram64v2_reduce_kernel sums.  The VALU form above reaches 13 TFLOP/s at f = 129 (LDS operand reads); this one is bound by
// the products (T tiles per four rows at 31.6 ns each per SIMD: tools/lab/src/f64_rates.hip).
typedef double f64x4_t __attribute__((ext_vector_type(4)));
template <int PW>
__global__ __launch_bounds__(512) void gram64m_kernel(const double* __restrict__ Y, int64_t m, int f, int bias, double* __restrict__ partial,
                                                      int64_t rows_per_block, int nb16, int ldy) {
    extern __shared__ __attribute__((aligned(16))) double sm64[];      // two images [16][ldy] of sixteen rows of Y~, zero padded
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int f4 = (f + 3) >> 2, nblk = f4 * (f4 + 1) / 2;
    const int ntile = nb16 * (nb16 + 1) / 2;
    int bi[PW], bj[PW];
    bool on[PW];
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        const int tl = wave + 8 * p;
        on[p] = tl < ntile;
        int a = 0, rem = on[p] ? tl : 0;
        while (rem >= nb16 - a) { rem -= nb16 - a; ++a; }           // (scalar: the tile number is wave-uniform)
        bi[p] = a; bj[p] = a + rem;
    }
    f64x4_t acc[PW];
#pragma unroll
    for (int p = 0; p < PW; ++p) acc[p] = f64x4_t{0.0, 0.0, 0.0, 0.0};
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(m, r0 + rows_per_block);
    constexpr int LD = 5;                                           // 16 x 144 doubles over 512 threads
    const int W = 16 * nb16;                                        // columns staged (ldy >= W)
    double stage[LD];
    auto fetch = [&](int64_t c0) {
#pragma unroll
        for (int k = 0; k < LD; ++k) {
            const int i = t + 512 * k, e = i / W, c = i - e * W;
            const int64_t row = c0 + e;
            const bool ok = e < 16 && row < r1 && c < f;
            const double v = Y[ok ? row * f + c : r0 * f];
            stage[k] = ok ? ((bias && c == 0) ? 1.0 : v) : 0.0;
        }
    };
    auto put = [&](int buf) {
#pragma unroll
        for (int k = 0; k < LD; ++k) {
            const int i = t + 512 * k, e = i / W, c = i - e * W;
            if (e < 16) sm64[(buf * 16 + e) * ldy + c] = stage[k];
        }
    };
    if (r0 < r1) { fetch(r0); put(0); }
    __syncthreads();
    int buf = 0;
    for (int64_t c0 = r0; c0 < r1; c0 += 16) {
        const bool more = c0 + 16 < r1;
        if (more) fetch(c0 + 16);
        const double* ys = sm64 + (buf * 16 + q) * ldy + r;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int p = 0; p < PW; ++p)
                acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(ys[4 * s * ldy + 16 * bi[p]], ys[4 * s * ldy + 16 * bj[p]], acc[p], 0, 0, 0);
        if (more) put(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    double* out = partial + (int64_t)blockIdx.x * nblk * 16;
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        if (!on[p]) continue;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int gi = 16 * bi[p] + q + 4 * v, gj = 16 * bj[p] + r;
            const int BI = gi >> 2, BJ = gj >> 2;
            if (BI < f4 && BJ < f4 && BJ >= BI) out[(int64_t)(BI * f4 - BI * (BI - 1) / 2 + (BJ - BI)) * 16 + 4 * (gi & 3) + (gj & 3)] = acc[p][v];
        }
    }
}

// G = sum of the partial blocks (fixed order: reproducible) + lambda I, both triangles
__global__ __launch_bounds__(256) void gram64v2_reduce_kernel(const double* __restrict__ partial, int nwg, int f, double lambda,
                                                              double* __restrict__ G) {
    const int f4 = (f + 3) >> 2, nblk = f4 * (f4 + 1) / 2;
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= nblk * 16) return;
    const int b = id >> 4, el = id & 15;
    int bi, bj;
    f64_decode(b, f4, false, bi, bj);
    const int gi = 4 * bi + (el >> 2), gj = 4 * bj + (el & 3);
    if (gi >= f || gj >= f || gj < gi) return;      // padding, or the mirrored half of a diagonal block
    double s = 0.0;
    for (int w = 0; w < nwg; ++w) s += partial[((int64_t)w * nblk + b) * 16 + el];
    if (gi == gj) s += lambda;
    G[(int64_t)gi * f + gj] = s;
    G[(int64_t)gj * f + gi] = s;
}

// ---- one row system per TEAM, in registers from the first gathered entry to the solution ---------------------------------
// TEAM = 256: the whole workgroup works on one row (barriers are __syncthreads).  TEAM = 64 (f <= 68: at most three blocks per
// lane): every WAVE of the workgroup has its own row and its own slice of LDS, and the steps below are ordered by the wave's
// lockstep execution (a wave-scope fence instead of a workgroup barrier) -- four rows in flight per workgroup where a narrow
// system would leave most of 256 threads without a block and every step waiting on a barrier.
// Dynamic LDS per team (doubles): ys [R][FP + 4] | wv [R] | pv [R] | panel [2][(f4 + 1) * 16] | dbuf [2][16] | yv [FP] | xs [4] |
// ib (int) [R] + flag
template <int TEAM> __device__ __forceinline__ void f64_team_sync() {
    if constexpr (TEAM == 256) {
        __syncthreads();
    } else {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}
static size_t solve64v2_team_doubles(int f, int R) {
    const int f4 = (f + 3) / 4, FP = 4 * f4;
    return (size_t)(R * (FP + 4) + 2 * R + 2 * (f4 + 1) * 16 + 32 + FP + 4) + (size_t)(R + 4 + 1) / 2 + 1;
}
// 1 / sqrt(x) to float64 accuracy from the hardware estimate (v_rsq_f64, ~2^-26) and two Newton steps; x > 0
__device__ __forceinline__ double f64_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double h = 0.5 * y, e = __builtin_fma(-x * y, y, 1.0);     // 1 - x y^2
        y = __builtin_fma(h, e, y);
    }
    return y;
}
// ---- the two phases every row system goes through, on the 4 x 4 blocks of a TEAM (solve64v2_kernel, solve64lr_kernel,
// factor64_kernel).  f4 = block rows / columns of the matrix; block column f4 is the right-hand side (RHS) or does not exist.
// Right-looking blocked Cholesky A = R^T R on the registers; with a right-hand side, yv = R^-T b when it returns.
template <int NB, int TEAM>
__device__ __forceinline__ bool f64_cholesky(double (&acc)[NB][16], const int (&bi)[NB], const int (&bj)[NB], const bool (&on)[NB], int f4,
                                             double* __restrict__ panel, double* __restrict__ dbuf, double* __restrict__ yv) {
    bool bad = false;
#pragma unroll 1
    for (int kb = 0; kb < f4; ++kb) {
        double* pan = panel + (kb & 1) * (f4 + 1) * 16;
        double* db = dbuf + (kb & 1) * 16;
#pragma unroll
        for (int n_ = 0; n_ < NB; ++n_) {
            if (on[n_] && bi[n_] == kb && bj[n_] == kb) {        // (a) the diagonal block: R_kk, then its inverse (upper triangular)
                double* a = acc[n_];
                double r[4][4], iv[4][4], rinv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    double dgl = a[5 * i];
#pragma unroll
                    for (int k = 0; k < i; ++k) dgl -= r[k][i] * r[k][i];
                    if (!(dgl > 0.0) || !(dgl < 1.0e300)) { bad = true; dgl = 1.0; }   // not positive definite (or not finite)
                    const double inv = f64_rsqrt(dgl);
                    rinv[i] = inv;
                    r[i][i] = dgl * inv;
#pragma unroll
                    for (int j = i + 1; j < 4; ++j) {
                        double v = a[4 * i + j];
#pragma unroll
                        for (int k = 0; k < i; ++k) v -= r[k][i] * r[k][j];
                        r[i][j] = v * inv;
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) iv[i][j] = 0.0;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    iv[j][j] = rinv[j];
#pragma unroll
                    for (int i = j - 1; i >= 0; --i) {
                        double v = 0.0;
#pragma unroll
                        for (int k = i + 1; k <= j; ++k) v += r[i][k] * iv[k][j];
                        iv[i][j] = -v * rinv[i];
                    }
                }
#pragma unroll
                for (int i = 0; i < 16; ++i) { a[i] = iv[i >> 2][i & 3]; db[i] = a[i]; }   // the owner keeps R_kk^-1 (back substitution; factor64: R^-1)
            }
        }
        f64_team_sync<TEAM>();
#pragma unroll
        for (int n_ = 0; n_ < NB; ++n_) {
            if (on[n_] && bi[n_] == kb && bj[n_] > kb) {         // (b) block row kb: R_kj = R_kk^-T A_kj
                double* a = acc[n_];
                double u[16];
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 4; ++y) {
                        double v = 0.0;
#pragma unroll
                        for (int k = 0; k <= x; ++k) v += db[4 * k + x] * a[4 * k + y];
                        u[4 * x + y] = v;
                    }
                double2* o = reinterpret_cast<double2*>(pan + bj[n_] * 16);
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = make_double2(u[2 * i], u[2 * i + 1]);
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = u[i];
                if (bj[n_] == f4) {                              // y_kb = (R^-T b)_kb: where the back substitution starts
#pragma unroll
                    for (int x = 0; x < 4; ++x) yv[4 * kb + x] = u[4 * x];
                }
            }
        }
        f64_team_sync<TEAM>();
#pragma unroll
        for (int n_ = 0; n_ < NB; ++n_) {
            if (on[n_] && bi[n_] > kb) {                         // (c) trailing update A_ij -= R_ki^T R_kj
                const double* ui = pan + bi[n_] * 16;
                const double* uj = pan + bj[n_] * 16;
                double* a = acc[n_];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const double2 i01 = *reinterpret_cast<const double2*>(ui + 4 * k), i23 = *reinterpret_cast<const double2*>(ui + 4 * k + 2);
                    const double2 j01 = *reinterpret_cast<const double2*>(uj + 4 * k), j23 = *reinterpret_cast<const double2*>(uj + 4 * k + 2);
                    const double iv_[4] = {i01.x, i01.y, i23.x, i23.y};
                    const double jv_[4] = {j01.x, j01.y, j23.x, j23.y};
#pragma unroll
                    for (int x = 0; x < 4; ++x)
#pragma unroll
                        for (int y = 0; y < 4; ++y) a[4 * x + y] = __builtin_fma(-iv_[x], jv_[y], a[4 * x + y]);
                }
            }
        }
    }
    return bad;
}

// Back substitution R x = yv, block column by block column; every R block is still in its owner's registers (diagonal blocks
// hold R_kk^-1).  x (nx values) goes to xout[i * xstride]; yv and xs are the team's LDS vectors.
template <int NB, int TEAM>
__device__ __forceinline__ void f64_backsub(const double (&acc)[NB][16], const int (&bi)[NB], const int (&bj)[NB], const bool (&on)[NB], int f4,
                                            double* __restrict__ yv, double* __restrict__ xs, double* __restrict__ xout, int nx) {
#pragma unroll 1
    for (int kb = f4 - 1; kb >= 0; --kb) {
#pragma unroll
        for (int n_ = 0; n_ < NB; ++n_) {
            if (on[n_] && bi[n_] == kb && bj[n_] == kb) {
                const double* iv = acc[n_];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    double v = 0.0;
#pragma unroll
                    for (int j = i; j < 4; ++j) v += iv[4 * i + j] * yv[4 * kb + j];
                    xs[i] = v;
                    if (4 * kb + i < nx) xout[4 * kb + i] = v;
                }
            }
        }
        f64_team_sync<TEAM>();
#pragma unroll
        for (int n_ = 0; n_ < NB; ++n_) {
            if (on[n_] && bj[n_] == kb && bi[n_] < kb) {         // y_bi -= R_(bi, kb) x_kb: one owner per (bi, kb)
                const double* u = acc[n_];
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    yv[4 * bi[n_] + x] -= u[4 * x] * xs[0] + u[4 * x + 1] * xs[1] + u[4 * x + 2] * xs[2] + u[4 * x + 3] * xs[3];
            }
        }
        f64_team_sync<TEAM>();
    }
}

#ifndef F64_WAVE_TEAM_OCC
#define F64_WAVE_TEAM_OCC 3      /* workgroups per CU the wave-team variants are compiled for (168 registers) */
#endif
template <int NB, int TEAM, int R>
__global__ __launch_bounds__(256, TEAM == 64 ? F64_WAVE_TEAM_OCC : 1) void solve64v2_kernel(const double* __restrict__ Y, int f, int bias, const double* __restrict__ G,
                                                        const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                        const double* __restrict__ vals, int64_t n, double* __restrict__ X,
                                                        int32_t* __restrict__ fb_rows, int32_t* __restrict__ fb_count, int team_doubles,
                                                        const int32_t* __restrict__ ctrl, const int32_t* __restrict__ state) {
    extern __shared__ __attribute__((aligned(16))) double sm64[];
    constexpr int NTEAM = 256 / TEAM;
    const bool lowrank_on = ctrl[0] != 0;                            // rows with 1 .. F64_LR_D entries go through solve64lr_kernel then
    const int t = threadIdx.x & (TEAM - 1);
    const int tw = TEAM == 256 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int f4 = (f + 3) >> 2, FP = 4 * f4, FPA = FP + 4, nblk = f4 * (f4 + 1) / 2 + f4;
    double* ys = sm64 + (size_t)tw * team_doubles;
    double* wv = ys + R * FPA;
    double* pv = wv + R;
    double* panel = pv + R;
    double* dbuf = panel + 2 * (f4 + 1) * 16;
    double* yv = dbuf + 32;
    double* xs = yv + FP;
    int* ib = reinterpret_cast<int*>(xs + 4);
    int* flag = ib + R;
    int bi[NB], bj[NB];
    bool on[NB];
#pragma unroll
    for (int n_ = 0; n_ < NB; ++n_) {
        const int b = n_ * TEAM + t;
        on[n_] = b < nblk;
        f64_decode(on[n_] ? b : 0, f4, true, bi[n_], bj[n_]);
    }
    for (int64_t row = (int64_t)blockIdx.x * NTEAM + tw; row < n; row += (int64_t)gridDim.x * NTEAM) {
        const int64_t lo = indptr[row], hi = indptr[row + 1];
        if (hi == lo) {                                              // no stored entries: zeros (wmf_model.py:274-276, :296-298)
            for (int c = t; c < f; c += TEAM) X[row * f + c] = 0.0;
            continue;
        }
        if (lowrank_on && hi - lo <= F64_LR_D) continue;
        if (lowrank_on && state[row]) continue;                      // solved by the matrix-free iteration (wmf_iter64.hip)
        double acc[NB][16];
#pragma unroll
        for (int n_ = 0; n_ < NB; ++n_)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[n_][i] = 0.0;
        if (t == 0) *flag = 0;
        // ---- A = sum_e w_e y_e y_e^T, b = sum_e (w_e + 1) y_e over the row's entries, R at a time
        for (int64_t c0 = lo; c0 < hi; c0 += R) {
            const int nvalid = (int)min((int64_t)R, hi - c0);
            f64_team_sync<TEAM>();                                   // the previous pass is done with ys / wv
            if (t < nvalid) {
                const int idx = indices[c0 + t];
                const double w = vals[c0 + t] - (bias ? Y[(int64_t)idx * f] : 0.0);   // data - bias[idx], :279
                ib[t] = idx; wv[t] = w; pv[t] = w + 1.0;
            }
            f64_team_sync<TEAM>();
            for (int i = t; i < nvalid * FPA; i += TEAM) {
                const int e = i / FPA, c = i - e * FPA;
                double v = 0.0;
                if (c < f) v = (bias && c == 0) ? 1.0 : Y[(int64_t)ib[e] * f + c];
                else if (c == FP) v = pv[e];
                ys[i] = v;
            }
            f64_team_sync<TEAM>();
            f64_accumulate<NB>(acc, bi, bj, on, ys, FPA, wv, nvalid, f4);
        }
        // ---- + G (lambda included); rows / columns of padding get a unit diagonal: their unknowns are zero
#pragma unroll
        for (int n_ = 0; n_ < NB; ++n_) {
            if (!on[n_] || bj[n_] == f4) continue;
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) {
                    const int gi = 4 * bi[n_] + x, gj = 4 * bj[n_] + y;
                    if (gi < f && gj < f) acc[n_][4 * x + y] += G[(int64_t)gi * f + gj];
                    else if (gi == gj) acc[n_][4 * x + y] += 1.0;
                }
        }
        // ---- blocked Cholesky A = R^T R on the registers (b rides along), then the back substitution
        const bool bad = f64_cholesky<NB, TEAM>(acc, bi, bj, on, f4, panel, dbuf, yv);
        bool give_up;
        if constexpr (TEAM == 256) {
            if (bad) *flag = 1;
            __syncthreads();
            give_up = *flag != 0;
            __syncthreads();                                         // (the flag is cleared again at the top of the next row)
        } else {
            give_up = __any(bad);
        }
        if (give_up) {                                               // uniform over the team: hand the row to the pivoted kernel
            if (t == 0) fb_rows[atomicAdd(fb_count, 1)] = (int32_t)row;
            continue;
        }
        f64_backsub<NB, TEAM>(acc, bi, bj, on, f4, yv, xs, X + row * f, f);
    }
}

// ---- the low-rank form in float64 for rows with few entries ---------------------------------------------------------------------
// The same push-through identity as the float32 path, in float64: with G = R^T R and V = Y~ R^-1,
//     x_u = R^-1 g_u,   g_u = V_u^T c,   c = E y,   (I + E S E) y = E^-1 p,   S = V_u V_u^T,  E = diag(sqrt(w)),  p = w + 1
// -- a d x d system (d <= 32: eight block columns instead of f / 4) built from ONE gather of d whitened rows.  Used when at
// least a quarter of the rows of a half step have 1 .. 32 entries (f64_decide_kernel); rows with a negative weight (bias
// models) or a system that is not positive definite go to the pivoted LU kernel like everywhere else.
//   factor64_kernel<NB>   G = R^T R by the blocked Cholesky above (one workgroup); blocks of R, diagonal blocks inverted
//   rinv64_kernel         R^-1 and R^-T, one thread per column, block back substitution
//   transform64_kernel    out = in~ . W (whitening V = Y~ R^-1; un-whitening X = g R^-T for the rows that took this path)
//   solve64lr_kernel      the row systems, one wave per row
#define F64_LR_RJ 16
__global__ void f64_count_low_kernel(const int64_t* __restrict__ indptr, int64_t n, int32_t* __restrict__ ctrl) {
    int c = 0;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t d = indptr[r + 1] - indptr[r];
        c += (d >= 1 && d <= F64_LR_D) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&ctrl[1], c);
}
// ctrl[0] = 1: the low-rank path is on for this half step (enough rows for the whitening pass to pay, debug flag not set)
// (iter_on: the matrix-free iteration of wmf_iter64.hip reads the whitened factors for rows of every length)
__global__ void f64_decide_kernel(int32_t* __restrict__ ctrl, int64_t n, int off, int iter_on) {
    ctrl[0] = (!off && n > 0 && (iter_on || (int64_t)ctrl[1] * 4 >= n)) ? 1 : 0;
}

template <int NB>
__global__ __launch_bounds__(256) void factor64_kernel(const double* __restrict__ G, int f, double* __restrict__ Rblk, int32_t* __restrict__ ctrl) {
    extern __shared__ __attribute__((aligned(16))) double sm64[];
    if (ctrl[0] == 0) return;
    const int t = threadIdx.x;
    const int f4 = (f + 3) >> 2, nblk = f4 * (f4 + 1) / 2;
    double* panel = sm64;                           // [2][(f4 + 1) * 16]
    double* dbuf = panel + 2 * (f4 + 1) * 16;       // [2][16]
    double* yv = dbuf + 32;                         // unused (no right-hand side)
    int* flag = reinterpret_cast<int*>(yv + 4);
    int bi[NB], bj[NB];
    bool on[NB];
    double acc[NB][16];
#pragma unroll
    for (int n_ = 0; n_ < NB; ++n_) {
        const int b = n_ * 256 + t;
        on[n_] = b < nblk;
        f64_decode(on[n_] ? b : 0, f4, false, bi[n_], bj[n_]);
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                const int gi = 4 * bi[n_] + x, gj = 4 * bj[n_] + y;
                acc[n_][4 * x + y] = (on[n_] && gi < f && gj < f) ? G[(int64_t)gi * f + gj] : ((gi == gj) ? 1.0 : 0.0);
            }
    }
    if (t == 0) *flag = 0;
    __syncthreads();
    const bool bad = f64_cholesky<NB, 256>(acc, bi, bj, on, f4, panel, dbuf, yv);
    if (bad) *flag = 1;
    __syncthreads();
    if (*flag) { if (t == 0) ctrl[0] = 0; return; }             // G + lambda I not positive definite: every row takes the direct kernel
#pragma unroll
    for (int n_ = 0; n_ < NB; ++n_) {
        if (!on[n_]) continue;
        double2* o = reinterpret_cast<double2*>(Rblk + ((int64_t)bi[n_] * f4 + bj[n_]) * 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = make_double2(acc[n_][2 * i], acc[n_][2 * i + 1]);
    }
}

// column j of R^-1 by block back substitution on the blocks of factor64_kernel (diagonal blocks hold R_kk^-1); written as
// Rinv[i][j] and RinvT[j][i], both [FP][FP], zero outside the f x f upper / lower triangle
__global__ __launch_bounds__(64) void rinv64_kernel(const double* __restrict__ Rblk, int f, double* __restrict__ Rinv, double* __restrict__ RinvT,
                                                    const int32_t* __restrict__ ctrl) {
    if (ctrl[0] == 0) return;
    const int f4 = (f + 3) >> 2, FP = 4 * f4;
    const int j = blockIdx.x * 64 + threadIdx.x;
    if (j >= FP) return;
    for (int i = 0; i < FP; ++i) { Rinv[(int64_t)i * FP + j] = 0.0; RinvT[(int64_t)j * FP + i] = 0.0; }
    if (j >= f) return;
    const int jb = j >> 2;
    for (int kb = jb; kb >= 0; --kb) {
        double rhs[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) rhs[x] = (4 * kb + x == j) ? 1.0 : 0.0;
        for (int lb = kb + 1; lb <= jb; ++lb) {
            const double* u = Rblk + ((int64_t)kb * f4 + lb) * 16;
            double xl[4];
#pragma unroll
            for (int y = 0; y < 4; ++y) xl[y] = Rinv[(int64_t)(4 * lb + y) * FP + j];          // this thread's own earlier results
#pragma unroll
            for (int x = 0; x < 4; ++x) rhs[x] -= u[4 * x] * xl[0] + u[4 * x + 1] * xl[1] + u[4 * x + 2] * xl[2] + u[4 * x + 3] * xl[3];
        }
        const double* iv = Rblk + ((int64_t)kb * f4 + kb) * 16;
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            double v = 0.0;
#pragma unroll
            for (int y = x; y < 4; ++y) v += iv[4 * x + y] * rhs[y];
            Rinv[(int64_t)(4 * kb + x) * FP + j] = v;
            RinvT[(int64_t)j * FP + 4 * kb + x] = v;
        }
    }
}

// out[r][0 .. f) = in~[r][0 .. f) . W, W [FP][FP] row-major (zero padded); in~ = in with column 0 read as 1 when set_col0_one.
// indptr != NULL: only the rows with 1 .. F64_LR_D stored entries are written (the un-whitening of the low-rank rows).
template <int NB>
__global__ __launch_bounds__(256) void transform64_kernel(const double* __restrict__ in, int64_t m, int f, const double* __restrict__ W,
                                                          int set_col0_one, double* __restrict__ out, const int64_t* __restrict__ indptr,
                                                          const int32_t* __restrict__ ctrl, int64_t rows_per_block, const int32_t* __restrict__ state) {
    extern __shared__ __attribute__((aligned(16))) double sm64[];
    if (ctrl[0] == 0) return;
    const int t = threadIdx.x;
    const int f4 = (f + 3) >> 2, FP = 4 * f4, LDR = FP + 1, nblk = 4 * f4;      // odd row stride: the four rows of a block on distinct banks
    double* ys = sm64;                              // [16][LDR]
    int rb[NB], cb[NB];
    bool on[NB];
#pragma unroll
    for (int n_ = 0; n_ < NB; ++n_) { const int b = n_ * 256 + t; on[n_] = b < nblk; rb[n_] = b & 3; cb[n_] = on[n_] ? b >> 2 : 0; }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(m, r0 + rows_per_block);
    for (int64_t c0 = r0; c0 < r1; c0 += 16) {
        const int nvalid = (int)min((int64_t)16, r1 - c0);
        __syncthreads();
        for (int i = t; i < 16 * FP; i += 256) {
            const int e = i / FP, c = i - e * FP;
            double v = 0.0;
            if (e < nvalid && c < f) v = (set_col0_one && c == 0) ? 1.0 : in[(c0 + e) * f + c];
            ys[e * LDR + c] = v;
        }
        __syncthreads();
#pragma unroll
        for (int n_ = 0; n_ < NB; ++n_) {
            if (!on[n_]) continue;
            double acc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.0;
            const double* a0 = ys + (4 * rb[n_]) * LDR;
            const double* wcol = W + 4 * cb[n_];
            for (int k = 0; k < f; ++k) {
                const double2 w01 = *reinterpret_cast<const double2*>(wcol + (int64_t)k * FP), w23 = *reinterpret_cast<const double2*>(wcol + (int64_t)k * FP + 2);
                const double w[4] = {w01.x, w01.y, w23.x, w23.y};
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    const double a = a0[x * LDR + k];
#pragma unroll
                    for (int y = 0; y < 4; ++y) acc[4 * x + y] = __builtin_fma(a, w[y], acc[4 * x + y]);
                }
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int e = 4 * rb[n_] + x;
                if (e >= nvalid) continue;
                const int64_t row = c0 + e;
                if (indptr) {                            // un-whitening: the low-rank rows and the rows the iteration solved
                    const int64_t d = indptr[row + 1] - indptr[row];
                    if ((d < 1 || d > F64_LR_D) && !(state && state[row])) continue;
                }
#pragma unroll
                for (int y = 0; y < 4; ++y) { const int c = 4 * cb[n_] + y; if (c < f) out[row * f + c] = acc[4 * x + y]; }
            }
        }
    }
}

// The same product on the matrix cores, for widths whose W fits LDS (FP . ldw doubles <= 160 KB: f <= 140): W is copied to LDS once per
// (persistent) workgroup, every wave then takes 16-row tiles on its own -- the tile's rows as the A operands of
// v_mfma_f64_16x16x4_f64 straight from global memory (lane (i, k) holds in[row0 + i][4 kk + k]: each row is read once, in
// 32-byte pieces), the 16-column blocks of W as B operands from LDS (row stride ldw = 16 mod 32 doubles: the four k of a read on
// disjoint banks), G column blocks at a time so that G accumulators are in flight.  The VALU form above spends 16 FMAs on 2
// global and 4 LDS loads per step (13 TFLOP/s at f = 129); here the tile costs f4 . nb products and nothing else.
template <int KK, int G>
__global__ __launch_bounds__(512) void transform64m_kernel(const double* __restrict__ in, int64_t m, int f, const double* __restrict__ W,
                                                           int set_col0_one, double* __restrict__ out, const int64_t* __restrict__ indptr,
                                                           const int32_t* __restrict__ ctrl, const int32_t* __restrict__ state, int ldw, int nb) {
    extern __shared__ __attribute__((aligned(16))) double sm64[];
    if (ctrl[0] == 0) return;
    const int f4 = (f + 3) >> 2, FP = 4 * f4;
    for (int i = threadIdx.x; i < FP * ldw; i += 512) {
        const int k = i / ldw, c = i - k * ldw;
        sm64[i] = c < FP ? W[(int64_t)k * FP + c] : 0.0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int64_t ntiles = (m + 15) >> 4;
    const int ngroups = (nb + G - 1) / G;
    for (int64_t tile = (int64_t)blockIdx.x * 8 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 8) {
        const int64_t row0 = tile << 4;
        // which of this lane's four output rows (row0 + q + 4 v) are written
        bool ok[4];
        bool any = false;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int64_t row = row0 + q + 4 * v;
            ok[v] = row < m;
            if (ok[v] && indptr) {                               // un-whitening: the low-rank rows and the rows the iteration solved
                const int64_t d = indptr[row + 1] - indptr[row];
                ok[v] = (d >= 1 && d <= F64_LR_D) || (state && state[row]);
            }
            any |= ok[v];
        }
        if (!__builtin_amdgcn_readfirstlane((int)(__ballot(any) != 0))) continue;
        const int64_t arow = row0 + r;
        const bool avalid = arow < m;
        const double* ap = in + (avalid ? arow : 0) * f;
        double a[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
            const int c = 4 * kk + q;
            a[kk] = (kk < f4 && c < f && avalid) ? ap[c] : 0.0;
        }
        if (set_col0_one && q == 0 && avalid) a[0] = 1.0;
        for (int g = 0; g < ngroups; ++g) {
            f64x4_t acc[G];
#pragma unroll
            for (int c = 0; c < G; ++c) acc[c] = f64x4_t{0.0, 0.0, 0.0, 0.0};
            const double* wb = sm64 + q * ldw + 16 * G * g + r;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) {                    // (no branch on kk < f4: a[kk] = 0 there and the read stays inside W)
                const double* wk = wb + 4 * (kk < f4 ? kk : 0) * ldw;
#pragma unroll
                for (int c = 0; c < G; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], wk[16 * c], acc[c], 0, 0, 0);
            }
#pragma unroll
            for (int c = 0; c < G; ++c) {
                const int col = 16 * (G * g + c) + r;
                if (col < f) {
#pragma unroll
                    for (int v = 0; v < 4; ++v)
                        if (ok[v]) out[(row0 + q + 4 * v) * f + col] = acc[c][v];
                }
            }
        }
    }
}

// one WAVE per row with 1 .. 32 stored entries.  LDS per wave (doubles): ys [16][36] | ones [16] | panel [2][9 * 16] | dbuf [32] | yv [32] |
// xs [4] | ev [32] | tv [32] | cv [32] | ib (int) [32]
#define F64_LR_TEAM_DOUBLES 1080
__global__ __launch_bounds__(256) void solve64lr_kernel(const double* __restrict__ V, const double* __restrict__ Y, int f, int bias,
                                                        const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                        const double* __restrict__ vals, int64_t n, double* __restrict__ gout,
                                                        int32_t* __restrict__ fb_rows, int32_t* __restrict__ fb_count,
                                                        const int32_t* __restrict__ ctrl, const int32_t* __restrict__ state) {
    extern __shared__ __attribute__((aligned(16))) double sm64[];
    if (ctrl[0] == 0) return;
    constexpr int F4MAX = F64_LR_D / 4, LDY = F64_LR_D + 4;
    const int t = threadIdx.x & 63;
    const int tw = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double* ys = sm64 + (size_t)tw * F64_LR_TEAM_DOUBLES;
    double* ones = ys + F64_LR_RJ * LDY;
    double* panel = ones + F64_LR_RJ;
    double* dbuf = panel + 2 * (F4MAX + 1) * 16;
    double* yv = dbuf + 32;
    double* xs = yv + F64_LR_D;
    double* ev = xs + 4;
    double* tv = ev + F64_LR_D;
    double* cv = tv + F64_LR_D;
    int* ib = reinterpret_cast<int*>(cv + F64_LR_D);
    int bi[1], bj[1];
    bool on[1];
    if (t < F64_LR_RJ) ones[t] = 1.0;
    for (int i = t; i < F64_LR_RJ * 4; i += 64) ys[(i >> 2) * LDY + F64_LR_D + (i & 3)] = 0.0;     // the right-hand-side columns stay zero
    for (int64_t row = (int64_t)blockIdx.x * 4 + tw; row < n; row += (int64_t)gridDim.x * 4) {
        const int64_t lo = indptr[row], hi = indptr[row + 1];
        const int d = (int)min(hi - lo, (int64_t)(F64_LR_D + 1));
        if (d < 1 || d > F64_LR_D) continue;
        if (state[row]) continue;                                    // solved by the matrix-free iteration (wmf_iter64.hip)
        // the system has f4 = ceil(d / 4) block rows (a 20-entry row: five block steps, not eight); its blocks are dealt to the lanes
        // for this row (the staged columns d .. 31, among them the right-hand-side block's, are zero)
        const int f4 = (d + 3) >> 2;
        on[0] = t < f4 * (f4 + 1) / 2 + f4;
        f64_decode(on[0] ? t : 0, f4, true, bi[0], bj[0]);
        f64_team_sync<64>();                                         // the previous row is done with the LDS vectors
        bool neg = false;
        if (t < F64_LR_D) {
            const bool real = t < d;
            const int idx = indices[real ? lo + t : lo];
            const double w = real ? vals[lo + t] - (bias ? Y[(int64_t)idx * f] : 0.0) : 1.0;      // data - bias[idx], :279
            if (real && !(w >= 0.0)) neg = true;                     // negative (or NaN) weight: E = sqrt(w) does not exist
            const double wc = w > 1.0e-300 ? w : 1.0e-300;            // a stored zero: p = 1 still counts (wmf_model.py:232,239)
            const double e = sqrt(wc);
            ib[t] = idx;
            ev[t] = real ? e : 1.0;
            tv[t] = real ? (w + 1.0) / e : 0.0;                      // E^-1 p
        }
        if (__any(neg)) {                                            // uniform: the pivoted kernel takes the row
            if (t == 0) fb_rows[atomicAdd(fb_count, 1)] = (int32_t)row;
            continue;
        }
        double acc[1][16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[0][i] = 0.0;
        // ---- S = V_u V_u^T over the features, F64_LR_RJ at a time (staged feature-major: ys[feature][entry])
        for (int j0 = 0; j0 < f; j0 += F64_LR_RJ) {
            const int nv = min(F64_LR_RJ, f - j0);
            f64_team_sync<64>();
            for (int i = t; i < F64_LR_RJ * F64_LR_D; i += 64) {
                const int e = i / F64_LR_RJ, jj = i - e * F64_LR_RJ;                 // consecutive lanes: consecutive features of one entry
                ys[jj * LDY + e] = (e < d && jj < nv) ? V[(int64_t)ib[e] * f + j0 + jj] : 0.0;
            }
            f64_team_sync<64>();
            f64_accumulate<1>(acc, bi, bj, on, ys, LDY, ones, nv, f4);
        }
        // ---- P = I + E S E, right-hand side E^-1 p
        if (on[0]) {
            if (bj[0] < f4) {
#pragma unroll
                for (int x = 0; x < 4; ++x)
#pragma unroll
                    for (int y = 0; y < 4; ++y) {
                        const int gi = 4 * bi[0] + x, gj = 4 * bj[0] + y;
                        acc[0][4 * x + y] = ev[gi] * acc[0][4 * x + y] * ev[gj] + (gi == gj ? 1.0 : 0.0);
                    }
            } else {
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    acc[0][4 * x] = tv[4 * bi[0] + x];
                    acc[0][4 * x + 1] = 0.0; acc[0][4 * x + 2] = 0.0; acc[0][4 * x + 3] = 0.0;
                }
            }
        }
        const bool bad = f64_cholesky<1, 64>(acc, bi, bj, on, f4, panel, dbuf, yv);
        if (__any(bad)) {
            if (t == 0) fb_rows[atomicAdd(fb_count, 1)] = (int32_t)row;
            continue;
        }
        f64_backsub<1, 64>(acc, bi, bj, on, f4, yv, xs, cv, F64_LR_D);          // y -> cv
        if (t < F64_LR_D) cv[t] = t < d ? ev[t] * cv[t] : 0.0;                    // c = E y
        f64_team_sync<64>();
        // ---- g = V_u^T c (the second read of the d whitened rows: L2)
        for (int j = t; j < f; j += 64) {
            double sacc = 0.0;
            for (int e = 0; e < d; ++e) sacc = __builtin_fma(V[(int64_t)ib[e] * f + j], cv[e], sacc);
            gout[row * f + j] = sacc;
        }
    }
}

// LU with partial pivoting for the rows of `rows` (count on the device): one workgroup per row (grid-stride); A (f x f), b (f)
// in this workgroup's workspace slice
__global__ __launch_bounds__(256) void solve64_lu_kernel(const double* __restrict__ Y, int f, int bias, const double* __restrict__ G,
                                                         const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                         const double* __restrict__ vals, const int32_t* __restrict__ rows,
                                                         const int32_t* __restrict__ count_ptr, double* __restrict__ X,
                                                         double* __restrict__ ws, int32_t* __restrict__ fail) {
    __shared__ double ys[WMF_MAX_F + 3];
    __shared__ double red_v[256];
    __shared__ int red_i[256];
    __shared__ int s_singular;
    const int t = threadIdx.x;
    double* A = ws + (size_t)blockIdx.x * ((size_t)f * f + f);
    double* b = A + (size_t)f * f;
    const int64_t n = *count_ptr;
    for (int64_t it = blockIdx.x; it < n; it += gridDim.x) {
        const int64_t row = rows[it];
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
    int64_t nb = (m + 255) / 256;                   // at least 256 rows (16 passes) per workgroup
    if (nb > 512) nb = 512;
    return (int)(nb < 1 ? 1 : nb);
}
static int solve64_blocks(int64_t n) { return (int)(n < 1 ? 1 : (n > 4096 ? 4096 : n)); }
#define WMF_F64_LU_GRID 256

// workspace: [gram partials nwg x blocks x 16][G f x f][LU slices WMF_F64_LU_GRID x (f x f + f)] doubles [64 int32: fallback count,
// control words][fallback rows n int32][R blocks f4 x f4 x 16][R^-1 FP x FP][R^-T FP x FP][V m x f][g n x f] doubles
static int64_t f64_al(int64_t bytes) { return (bytes + 255) / 256 * 256; }
int64_t wmf_f64_ws_bytes(int f, int64_t m, int64_t n) {
    const int64_t ff = (int64_t)f * f, f4 = (f + 3) / 4, FP = 4 * f4;
    return f64_al(8 * ((int64_t)gram64_blocks(m) * f64_blocks(f, false) * 16 + ff + (int64_t)WMF_F64_LU_GRID * (ff + f))) + f64_al(4 * (n + 64)) +
           f64_al(8 * f4 * f4 * 16) + 2 * f64_al(8 * FP * FP) + f64_al(8 * m * f) + f64_al(8 * (n > 0 ? n : 1) * f) + f64_al(4 * (n > 0 ? n : 1)) + 256;
}

template <int NB>
static void launch_gram64(const double* Y, int64_t m, int f, int bias, double lambda, double* partial, double* G, int nwg, hipStream_t st) {
    const int f4 = (f + 3) / 4, FP = 4 * f4;
    const size_t lds_g = (size_t)(F64_R * FP + F64_R) * 8;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)gram64v2_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        attr_set = true;
    }
    static const char* nmg = wmf_kname("gram64v2_kernel<%d>", NB);
    const int64_t rpb = (m + nwg - 1) / nwg;
    const int nb16 = (f + 15) / 16, pw = (nb16 * (nb16 + 1) / 2 + 7) / 8, ldy = 16 * nb16 + ((nb16 & 1) ? 0 : 16);
    if (pw <= 6 && !(wmf_debug_flags & 536870912)) {        // the matrix-core form (f <= 144)
#define GM_(PW)                                                                                                            \
    case PW: {                                                                                                             \
        static const char* nm_ = wmf_kname("gram64m_kernel<%d>", PW);                                                      \
        WMF_LAUNCH(nm_, (gram64m_kernel<PW>), dim3(nwg), dim3(512), (size_t)2 * 16 * ldy * 8, st, Y, m, f, bias, partial, rpb, nb16, ldy); \
    } break;
        switch (pw) { GM_(1) GM_(2) GM_(3) GM_(4) GM_(5) GM_(6) }
#undef GM_
    } else
    WMF_LAUNCH(nmg, (gram64v2_kernel<NB>), dim3(nwg), dim3(256), lds_g, st, Y, m, f, bias, partial, rpb);
    const int nel = f64_blocks(f, false) * 16;
    WMF_LAUNCH("gram64v2_reduce_kernel", gram64v2_reduce_kernel, dim3((unsigned)((nel + 255) / 256)), dim3(256), 0, st, partial, nwg, f,
               lambda, G);
}

template <int NB, int TEAM, int R>
static void launch_solve64(const double* Y, int f, int bias, const double* G, const int64_t* indptr, const int32_t* indices,
                           const double* values, int64_t n, double* X, int32_t* fb_rows, int32_t* fb_count, const int32_t* ctrl, hipStream_t st,
                           const int32_t* state) {
    const int team_doubles = (int)((solve64v2_team_doubles(f, R) + 1) & ~(size_t)1);           // 16-byte aligned slices
    const size_t lds = (size_t)team_doubles * 8 * (256 / TEAM);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)solve64v2_kernel<NB, TEAM, R>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        attr_set = true;
    }
    static const char* nms = wmf_kname("solve64v2_kernel<%d, %d, %d>", NB, TEAM, R);
    const int64_t teams = 256 / TEAM;
    int64_t grid = (n + teams - 1) / teams;
    if (grid > 4096) grid = 4096;
    WMF_LAUNCH(nms, (solve64v2_kernel<NB, TEAM, R>), dim3((unsigned)grid), dim3(256), lds, st, Y, f, bias, G, indptr, indices, values, n,
               X, fb_rows, fb_count, team_doubles, ctrl, state);
}

template <int NB>
static void launch_factor64(const double* G, int f, double* Rblk, int32_t* ctrl, hipStream_t st) {
    const int f4 = (f + 3) / 4;
    static const char* nm = wmf_kname("factor64_kernel<%d>", NB);
    WMF_LAUNCH(nm, (factor64_kernel<NB>), dim3(1), dim3(256), (size_t)(2 * (f4 + 1) * 16 + 32 + 8) * 8, st, G, f, Rblk, ctrl);
}

template <int NB>
static void launch_transform64(const double* in, int64_t m, int f, const double* W, int set_col0_one, double* out, const int64_t* indptr,
                               const int32_t* ctrl, hipStream_t st, const int32_t* state = nullptr) {
    if (m <= 0) return;
    const int f4 = (f + 3) / 4, FP = 4 * f4;
    {                                               // the matrix-core form where W fits LDS (debug flag 536870912: never)
        const int nb = (f + 15) / 16;
        const int ldw = 16 * nb + ((nb & 1) ? 0 : 16);          // 16 mod 32
        const size_t lds = (size_t)FP * ldw * 8;
        if (lds <= 160 * 1024 && f4 <= 36 && !(wmf_debug_flags & 536870912)) {
            int64_t grid = ((m + 15) / 16 + 7) / 8;
            if (grid > 256) grid = 256;
#define TM_(KK, G)                                                                                                                     \
    do {                                                                                                                               \
        static bool set_ = false;                                                                                                      \
        if (!set_) {                                                                                                                   \
            (void)hipFuncSetAttribute((const void*)transform64m_kernel<KK, G>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
            set_ = true;                                                                                                               \
        }                                                                                                                              \
        static const char* nm_ = wmf_kname("transform64m_kernel<%d, %d>", KK, G);                                                      \
        WMF_LAUNCH(nm_, (transform64m_kernel<KK, G>), dim3((unsigned)grid), dim3(512), lds, st, in, m, f, W, set_col0_one, out, indptr, \
                   ctrl, state, ldw, nb);                                                                                              \
    } while (0)
            if (f4 <= 16) TM_(16, 2);
            else if (f4 == 17) TM_(17, 3);
            else if (f4 <= 20) TM_(20, 3);
            else if (f4 <= 24) TM_(24, 3);
            else if (f4 <= 28) TM_(28, 3);
            else if (f4 <= 32) TM_(32, 3);
            else if (f4 == 33) TM_(33, 3);
            else TM_(36, 3);
#undef TM_
            return;
        }
    }
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)transform64_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        attr_set = true;
    }
    int64_t grid = (m + 63) / 64;                   // at least four 16-row passes per workgroup
    if (grid > 2048) grid = 2048;
    const int64_t rpb = ((m + grid - 1) / grid + 15) / 16 * 16;
    static const char* nm = wmf_kname("transform64_kernel<%d>", NB);
    WMF_LAUNCH(nm, (transform64_kernel<NB>), dim3((unsigned)((m + rpb - 1) / rpb)), dim3(256), (size_t)16 * (FP + 1) * 8, st, in, m, f, W,
               set_col0_one, out, indptr, ctrl, rpb, state);
}

int wmf_launch_half_step_f64(const double* Y, int64_t m, int f, int bias, const int64_t* indptr, const int32_t* indices,
                             const double* values, int64_t n, double lambda, double* X, void* ws, int32_t* fail, hipStream_t st) {
    const int64_t ff = (int64_t)f * f, f4 = (f + 3) / 4, FP = 4 * f4;
    const int nwg = gram64_blocks(m);
    char* base = static_cast<char*>(ws);
    double* partial = reinterpret_cast<double*>(base);
    double* G = partial + (int64_t)nwg * f64_blocks(f, false) * 16;
    double* slices = G + ff;
    base += f64_al(8 * ((int64_t)nwg * f64_blocks(f, false) * 16 + ff + (int64_t)WMF_F64_LU_GRID * (ff + f)));
    int32_t* fb_count = reinterpret_cast<int32_t*>(base);
    int32_t* ctrl = fb_count + 16;                  // [0] low-rank path on, [1] rows with 1 .. F64_LR_D entries
    int32_t* fb_rows = fb_count + 64;
    base += f64_al(4 * (n + 64));
    double* Rblk = reinterpret_cast<double*>(base);
    base += f64_al(8 * f4 * f4 * 16);
    double* Rinv = reinterpret_cast<double*>(base);
    base += f64_al(8 * FP * FP);
    double* RinvT = reinterpret_cast<double*>(base);
    base += f64_al(8 * FP * FP);
    double* V = reinterpret_cast<double*>(base);
    base += f64_al(8 * m * f);
    double* gbuf = reinterpret_cast<double*>(base);
    base += f64_al(8 * (n > 0 ? n : 1) * f);
    int32_t* state = reinterpret_cast<int32_t*>(base);          // [n] 1: the row was solved by the matrix-free iteration (wmf_iter64.hip)
    if (hipMemsetAsync(fb_count, 0, 256, st) != hipSuccess) return -2;
    if (n > 0 && hipMemsetAsync(state, 0, (size_t)n * 4, st) != hipSuccess) return -2;
    // round 4: rows whose whitened system is close to the identity by a matrix-free Neumann series in float64 (wmf_iter64.hip)
    // -- first: what it marks solved, the two kernels below skip; debug flags 134217728 (no whitened path) / 268435456 switch it off
    const bool iter_on = wmf_iter_enabled() && wmf_iter64_dmax(f) > 0 && !(wmf_debug_flags & 134217728);
    switch (f64_nb(f64_blocks(f, false))) {
#define C_(N) case N: launch_gram64<N>(Y, m, f, bias, lambda, partial, G, nwg, st); break;
        C_(1) C_(2) C_(3) C_(5) C_(9)
#undef C_
        default: return -1;
    }
    if (n > 0) {
        // ---- rows with 1 .. 32 entries through the whitened low-rank form, when there are enough of them (debug flag 134217728: never)
        int64_t cgrid = (n + 255) / 256;
        if (cgrid > 1024) cgrid = 1024;
        hipLaunchKernelGGL(f64_count_low_kernel, dim3((unsigned)cgrid), dim3(256), 0, st, indptr, n, ctrl);
        hipLaunchKernelGGL(f64_decide_kernel, dim3(1), dim3(1), 0, st, ctrl, n, (wmf_debug_flags & 134217728) ? 1 : 0, iter_on ? 1 : 0);
        switch (f64_nb(f64_blocks(f, false))) {
#define C_(N) case N: launch_factor64<N>(G, f, Rblk, ctrl, st); break;
            C_(1) C_(2) C_(3) C_(5) C_(9)
#undef C_
            default: return -1;
        }
        WMF_LAUNCH("rinv64_kernel", rinv64_kernel, dim3((unsigned)((FP + 63) / 64)), dim3(64), 0, st, Rblk, f, Rinv, RinvT, ctrl);
        if (4 * f4 <= 256) launch_transform64<1>(Y, m, f, Rinv, bias, V, nullptr, ctrl, st);
        else launch_transform64<2>(Y, m, f, Rinv, bias, V, nullptr, ctrl, st);
        if (iter_on) {
            if (wmf_launch_iter64(V, Y, f, bias, indptr, indices, values, n, 1, gbuf, state, ctrl, st)) return -1;
            if (wmf_launch_iter64(V, Y, f, bias, indptr, indices, values, n, 0, gbuf, state, ctrl, st)) return -1;
        }
        {
            static bool attr_set = false;
            if (!attr_set) {
                (void)hipFuncSetAttribute((const void*)solve64lr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
                attr_set = true;
            }
            int64_t grid = (n + 3) / 4;
            if (grid > 4096) grid = 4096;
            WMF_LAUNCH("solve64lr_kernel", solve64lr_kernel, dim3((unsigned)grid), dim3(256), (size_t)4 * F64_LR_TEAM_DOUBLES * 8, st, V, Y, f, bias,
                       indptr, indices, values, n, gbuf, fb_rows, fb_count, ctrl, state);
        }
        if (4 * f4 <= 256) launch_transform64<1>(gbuf, n, f, RinvT, 0, X, indptr, ctrl, st, state);
        else launch_transform64<2>(gbuf, n, f, RinvT, 0, X, indptr, ctrl, st, state);
        // ---- every other row (all of them when the low-rank path is off): the f x f system directly
        const int nblk = f64_blocks(f, true);
#define S_(N, T, R) launch_solve64<N, T, R>(Y, f, bias, G, indptr, indices, values, n, X, fb_rows, fb_count, ctrl, st, state)
        const bool waves = !(wmf_debug_flags & 67108864);       // debug flag 67108864 (timing experiments): workgroup teams at every width
        if (!waves && nblk <= 256) S_(1, 256, 16);
        else if (nblk <= 64) S_(1, 64, 8);            // one WAVE per row while a lane holds at most three blocks (f <= 68)
        else if (nblk <= 128) S_(2, 64, 8);
        else if (nblk <= 192) S_(3, 64, 8);
        else if (nblk <= 256) S_(1, 256, 16);         // one workgroup per row beyond
        else if (nblk <= 512) S_(2, 256, 16);
        else if (nblk <= 768) S_(3, 256, 16);
        else if (nblk <= 1280) S_(5, 256, 16);
        else S_(9, 256, 16);
#undef S_
        // rows neither kernel could take (negative weights, not positive definite; count on the device; none as a rule)
        WMF_LAUNCH("solve64_lu_kernel", solve64_lu_kernel, dim3(WMF_F64_LU_GRID), dim3(256), 0, st, Y, f, bias, G, indptr, indices,
                   values, fb_rows, fb_count, X, slices, fail);
    }
    return 0;
}

int wmf_launch_confidence_f64(double* values, int64_t nnz, double alpha, double beta, int mode, hipStream_t st) {
    if (nnz <= 0) return 0;
    int64_t grid = (nnz + 255) / 256;
    if (grid > 4096) grid = 4096;
    WMF_LAUNCH("confidence64_kernel", confidence64_kernel, dim3((unsigned)grid), dim3(256), 0, st, values, nnz, alpha, beta, mode);
    return 0;
}
