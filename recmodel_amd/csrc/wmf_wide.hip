// Rows with more than 32 stored entries when 144 < f <= 272 (k = 256, with or without biases): the
// f x f whitened system  (I + V_u^T D V_u) g = V_u^T p  no longer fits the accumulator registers of one wave
// (wmf_directw.hip), so ONE 512-THREAD WORKGROUP takes a row and its eight waves share the 16 x 16 tiles.
//
// Reference arithmetic: RecModel/wmf_model.py:233-239 (per-row Gramian + np.linalg.solve); SPD whenever the
// weights are non-negative, so a block LDL^T elimination replaces LU; a row that is not positive definite
// (a non-positive pivot) is bounced to the pivoted LU kernel below.
//
// Tiles: (bi, bj), bi <= bj <= NFB, enumerated row by row; bj = NFB is the right hand side riding along as one
// more block column.  Tile t belongs to wave t mod 8, accumulator slot t div 8.  Unlike wmf_direct.hip the
// eight waves run THE SAME code: a slot's (bi, bj) are wave-uniform run-time values kept in SGPRs, every MFMA
// operand comes from LDS at a run-time offset, and the block-row loop of the factorisation is a run-time loop
// whose body tests each slot against the pivot.  Register indices stay compile-time constants (the slot
// number) without eight specialised copies of the code -- which at NFB = 17 cost more registers than they
// saved (440 spills with the wmf_direct.hip scheme, none here).
//   A. entries staged 32 at a time through LDS (two register sets in flight, the next row's first chunks are
//      requested before this row's factorisation):  tile(bi, bj) += frag[bi]^T (w frag[bj]).
//   C. block elimination without square roots, as in wmf_directw.hip: for p = 0 .. NFB-1 the wave that owns tile
//      (p,p) inverts it in its registers (16 DPP Gauss-Jordan steps) and publishes X; W_pj = X B_pj with the
//      tile's own registers as the B operand; originals and W to two LDS panels; B_ij -= B_pi^T W_pj.
//   D. g_p = w_p - sum_{j>p} W_pj g_j, column by column; every element of the running right hand side has exactly
//      one writer per step, so the summation order is fixed (no atomics).
#include "wmf_common.h"
#include "wmf_internal.h"

template <int NFB>
struct WideCfg {
    static constexpr int NW = 8, NTHR = 64 * NW;
    static constexpr int FP = 16 * NFB;
    static constexpr int LDV = (FP % 32 == 16) ? FP : FP + 16;   // = 16 (mod 32): the two k rows of a half wave hit disjoint banks
    static constexpr int RC = 32;
    static constexpr int NT = NFB * (NFB + 1) / 2 + NFB;          // upper tiles + one rhs tile per block row
    static constexpr int NACC = (NT + NW - 1) / NW;
    static constexpr int PF = (RC * (FP / 4) + NTHR - 1) / NTHR;  // 16-byte pieces prefetched per thread
    // LDS carve (floats)
    static constexpr int OFF_VS = 0;
    static constexpr int OFF_W = OFF_VS + RC * LDV + 16;          // 16 spare floats: the rhs slot's unused B address stays inside
    static constexpr int OFF_P = OFF_W + RC;
    static constexpr int OFF_T = OFF_P + RC;                       // [NFB][16][20] inverses of the pivot tiles
    static constexpr int OFF_PAN = OFF_T + NFB * 320;              // [NFB + 1][16][20] original tiles of block row p
    static constexpr int OFF_PAN2 = OFF_PAN + (NFB + 1) * 320;     // [NFB + 1][16][20] W tiles of block row p
    static constexpr int OFF_Z = OFF_PAN2 + (NFB + 1) * 320;       // [FP] running rhs of the backward pass, then the solution
    static constexpr int OFF_FLAG = OFF_Z + FP;                    // [4]
    static constexpr int TOTAL = OFF_FLAG + 4;
};

template <int NFB>
__global__ __launch_bounds__(512, 1) void solve_wide_kernel(const int32_t* __restrict__ rows, int64_t count,
                                                            const float* __restrict__ V, const float* __restrict__ biasv,
                                                            const int64_t* __restrict__ indptr,
                                                            const int32_t* __restrict__ indices,
                                                            const float* __restrict__ vals, int f, int ld,
                                                            float* __restrict__ g, int32_t* __restrict__ fb_rows,
                                                            int32_t* __restrict__ fb_count, int dbg, const int32_t* __restrict__ count_dev) {
    if (count_dev) count = *count_dev;                           // (the rows the iteration kernel bounced: the count is on the device)
    using C = WideCfg<NFB>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sm = reinterpret_cast<float*>(smem_raw);
    float* Vs = sm + C::OFF_VS; float* wsm = sm + C::OFF_W; float* psm = sm + C::OFF_P;
    float* T = sm + C::OFF_T; float* Pan = sm + C::OFF_PAN; float* Pan2 = sm + C::OFF_PAN2;
    float* zb = sm + C::OFF_Z; int* flag = reinterpret_cast<int*>(sm + C::OFF_FLAG);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int nch = ld >> 2;

    // this wave's tiles: slot a <-> tile t = 8 a + wave; (-1, -1) for a slot past the last tile
    int ti[C::NACC], tj[C::NACC];
#pragma unroll
    for (int a = 0; a < C::NACC; ++a) {
        int rem = a * C::NW + wave, bi = 0;
        while (bi < NFB && rem >= NFB + 1 - bi) { rem -= NFB + 1 - bi; ++bi; }
        ti[a] = __builtin_amdgcn_readfirstlane(bi < NFB ? bi : -1);
        tj[a] = __builtin_amdgcn_readfirstlane(bi < NFB ? bi + rem : -1);
    }

    // per-thread (row-in-chunk, piece) of each prefetched 16-byte piece; fixed across chunks
    int pj[C::PF], pc[C::PF];
#pragma unroll
    for (int i = 0; i < C::PF; ++i) { const int e = tid + C::NTHR * i; pj[i] = e / nch; pc[i] = e % nch; }

    float4 preA[C::PF], preB[C::PF];
    float wA = 0.f, wB = 0.f;
    auto load_chunk = [&](float4 (&pre)[C::PF], float& wpre, int64_t lo_, int d_, int base) {
        const int nrow = min(C::RC, d_ - base);                  // may be <= 0: everything masked
#pragma unroll
        for (int i = 0; i < C::PF; ++i) {                        // unconditional loads, masked by multiplication
            const float on = pj[i] < nrow ? 1.f : 0.f;
            const int idx = indices[pj[i] < nrow ? lo_ + base + pj[i] : 0];
            const float4 v = reinterpret_cast<const float4*>(V + (int64_t)idx * ld)[pc[i]];
            pre[i] = make_float4(v.x * on, v.y * on, v.z * on, v.w * on);
        }
        {
            const bool on = tid < nrow;
            const int64_t e = on ? lo_ + base + tid : 0;
            float wv = vals[e];
            if (biasv) wv -= biasv[indices[e]];
            wpre = on ? wv : 0.f;
        }
    };
    for (int e = tid; e < C::RC * C::LDV + 16; e += C::NTHR) Vs[e] = 0.f;   // pad columns [ld, LDV) stay zero for good

    int64_t it = blockIdx.x;
    int u = 0, d = 0;
    int64_t lo = 0;
    if (it < count) {
        u = rows[it]; lo = indptr[u]; d = (int)(indptr[u + 1] - lo);
        load_chunk(preA, wA, lo, d, 0);
        load_chunk(preB, wB, lo, d, C::RC);
    }
    for (; it < count; it += gridDim.x) {
        const int nchunks = (d + C::RC - 1) / C::RC;
        const int64_t itn = it + gridDim.x;
        int un = 0, dn = 0;
        int64_t lon = 0;
        if (itn < count) { un = rows[itn]; lon = indptr[un]; dn = (int)(indptr[un + 1] - lon); }

        f32x4 acc[C::NACC];
#pragma unroll
        for (int a = 0; a < C::NACC; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tid == 0) flag[0] = 0;

        // ---- A. slot 0 = set A (even chunks), slot 1 = set B (odd chunks); the next row's chunk `slot` goes into
        //      the same set once this row no longer needs it
        auto consume = [&](float4 (&pre)[C::PF], float& wpre, int c, int slot) {
            const int base = c * C::RC;
            const int nrow = min(C::RC, d - base);
            __syncthreads();                                     // everyone finished reading the previous chunk
#pragma unroll
            for (int i = 0; i < C::PF; ++i)
                if (pj[i] < C::RC) *reinterpret_cast<float4*>(&Vs[pj[i] * C::LDV + 4 * pc[i]]) = pre[i];   // zeros beyond nrow
            if (tid < C::RC) { wsm[tid] = wpre; psm[tid] = (tid < nrow) ? wpre + 1.f : 0.f; }
            __syncthreads();
            if (c + 2 < nchunks) load_chunk(pre, wpre, lo, d, base + 2 * C::RC);
            else if (itn < count) load_chunk(pre, wpre, lon, dn, slot * C::RC);
            const int nsteps = WMF_ABL(dbg, 2) ? 0 : (nrow + 3) >> 2;           // dbg: timing ablations (wmf_debug_set_flags)
            for (int s = 0; s < nsteps; ++s) {
                const float wq = wsm[4 * s + q];
                const float pb = (r == 0) ? psm[4 * s + q] : 0.f;          // rhs tile: p in column 0
                const float* vrow = Vs + (4 * s + q) * C::LDV + r;
#pragma unroll
                for (int a = 0; a < C::NACC; ++a) {
                    if (a < C::NACC - 1 || ti[a] >= 0) {                    // only the last slot can be empty
                        const float av = vrow[16 * ti[a]];
                        const float bv = vrow[16 * tj[a]];                  // bj = NFB reads the spare columns; unused then
                        acc[a] = WMF_MFMA16(av, tj[a] < NFB ? bv * wq : pb, acc[a]);
                    }
                }
            }
        };
        for (int c = 0; c < nchunks; c += 2) {
            consume(preA, wA, c, 0);
            if (c + 1 < nchunks) consume(preB, wB, c + 1, 1);
        }
        if (itn < count) {                                       // sets this row never consumed
            if (nchunks < 1) load_chunk(preA, wA, lon, dn, 0);
            if (nchunks < 2) load_chunk(preB, wB, lon, dn, C::RC);
        }

        // ---- C. block elimination, tiles in registers
#pragma unroll
        for (int a = 0; a < C::NACC; ++a) {
            if (ti[a] >= 0 && ti[a] == tj[a]) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) if (r == 4 * q + reg) acc[a][reg] += 1.f;
            }
        }
        int baddr[4];
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) baddr[kq] = (r + 16 * kq) * 4;
#pragma unroll 1
        for (int p = 0; p < (WMF_ABL(dbg, 1) ? 0 : NFB); ++p) {
            // (a) the wave that owns tile (p, p) inverts it in registers: a symmetric tile in accumulator layout is
            //     the row-distributed layout of the Gauss-Jordan sweep (wmf_common.h), 16 DPP steps, no other wave waits
            //     on a serial Cholesky.  X goes to T[p] as [row][col] for everybody's A operand.
#pragma unroll
            for (int a = 0; a < C::NACC; ++a) {
                if (ti[a] == p && tj[a] == p) {
                    f32x4 X = acc[a];
                    bool ok = true;
                    gj_inv_sweep<true, true, true>(X, baddr, r, q, ok, std::make_integer_sequence<int, 16>{});   // (pivots above WMF_PIVOT_CAP bounce the row)
                    if (!ok && lane == 0) flag[0] = 1;
                    *reinterpret_cast<float4*>(T + p * 320 + r * 20 + 4 * q) = make_float4(X[0], X[1], X[2], X[3]);
                }
            }
            __syncthreads();
            {                                                    // (c) row panel: W_pj = X B_pj (the tile itself is the B operand);
                                                                 //     originals to Pan1, W to Pan2, W stays in the registers
                const float4 x4 = *reinterpret_cast<const float4*>(T + p * 320 + r * 20 + 4 * q);
#pragma unroll
                for (int a = 0; a < C::NACC; ++a) {
                    if (ti[a] == p && tj[a] > p) {
                        f32x4 n = f32x4{0.f, 0.f, 0.f, 0.f};
                        n = WMF_MFMA16(x4.x, acc[a][0], n); n = WMF_MFMA16(x4.y, acc[a][1], n);
                        n = WMF_MFMA16(x4.z, acc[a][2], n); n = WMF_MFMA16(x4.w, acc[a][3], n);
                        float* d1 = Pan + tj[a] * 320 + (4 * q) * 20 + r;
                        float* d2 = Pan2 + tj[a] * 320 + (4 * q) * 20 + r;
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) { d1[reg * 20] = acc[a][reg]; d2[reg * 20] = n[reg]; }
                        acc[a] = n;
                    }
                }
            }
            __syncthreads();
#pragma unroll
            for (int a = 0; a < C::NACC; ++a) {                  // (e) trailing update  B_ij -= B_pi^T W_pj,  p < i <= j <= NFB
                if (ti[a] > p) {
                    const float* pa = Pan + ti[a] * 320 + (4 * q) * 20 + r;
                    const float* pb = Pan2 + tj[a] * 320 + (4 * q) * 20 + r;
                    f32x4 c = acc[a];
#pragma unroll
                    for (int e = 0; e < 4; ++e) c = WMF_MFMA16(-pa[e * 20], pb[e * 20], c);
                    acc[a] = c;
                }
            }
            __syncthreads();
        }
        // w_p = column 0 of tile (p, NFB) (now W_p,rhs = X_p y_p)
#pragma unroll
        for (int a = 0; a < C::NACC; ++a) {
            if (ti[a] >= 0 && tj[a] == NFB && r == 0) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) zb[16 * ti[a] + 4 * q + reg] = acc[a][reg];
            }
        }
        __syncthreads();
        // ---- D. g_p = w_p - sum_{j > p} W_pj g_j, column by column: once g_p is final every tile (i, p), i < p, takes
        //      its product out of z_i -- one tile, hence one writer, per block i and step
#pragma unroll 1
        for (int p = (WMF_ABL(dbg, 1) ? -1 : NFB - 1); p >= 0; --p) {
            const float gp = zb[16 * p + r];                     // final: all columns j > p have been taken out
#pragma unroll
            for (int a = 0; a < C::NACC; ++a) {
                if (tj[a] == p && ti[a] < p && ti[a] >= 0) {
                    float v0 = acc[a][0] * gp, v1 = acc[a][1] * gp, v2 = acc[a][2] * gp, v3 = acc[a][3] * gp;
                    wmf_row16_sum4(v0, v1, v2, v3);              // sums over the 16 columns of the tile
                    if (r == 0) {
                        float* z = zb + 16 * ti[a] + 4 * q;
                        z[0] -= v0; z[1] -= v1; z[2] -= v2; z[3] -= v3;
                    }
                }
            }
            __syncthreads();
        }
        float* gs = zb;                                          // the solution is what is left in z
        const bool notpd = flag[0] != 0;
        if (notpd) {
            if (tid == 0) fb_rows[atomicAdd(fb_count, 1)] = u;   // not positive definite: the LU kernel redoes it
        } else {
            for (int c = tid; c < ld; c += C::NTHR) g[(int64_t)u * ld + c] = (c < f) ? gs[c] : 0.f;
        }
        u = un; lo = lon; d = dn;
        __syncthreads();                                         // gs / flag are reused by the next row
    }
}

// -------------------------------------------------------------------------- pivoted LU, wide f
// Fallback for rows with negative weights or a system that is not positive definite when f > 144 (the LDS
// version in wmf_solve.hip holds f <= 144).  One workgroup per row; the f x (f + 1) augmented matrix lives in
// a global workspace slice of this workgroup (L2 resident).  LAPACK gesv order: partial pivoting, first
// largest |entry| wins.  Rare path: correctness first.
__global__ __launch_bounds__(256) void solve_wide_lu_kernel(const int32_t* __restrict__ rows,
                                                            const int32_t* __restrict__ count_ptr,
                                                            const float* __restrict__ V, const float* __restrict__ biasv,
                                                            const int64_t* __restrict__ indptr,
                                                            const int32_t* __restrict__ indices,
                                                            const float* __restrict__ vals, int f, int ld,
                                                            float* __restrict__ g, int32_t* __restrict__ fail_count,
                                                            float* __restrict__ work) {
    constexpr int RC = 16, LDS_LD = 276;
    __shared__ __attribute__((aligned(16))) float Vs[RC * LDS_LD];
    __shared__ float wsm[RC];
    __shared__ float red[8];
    const int tid = threadIdx.x;
    const int LDB = f + 1;                                      // column f = right hand side
    float* B = work + (size_t)blockIdx.x * f * LDB;
    const int nch = ld >> 2;
    const int64_t total = *count_ptr;
    for (int64_t it = blockIdx.x; it < total; it += gridDim.x) {
        const int u = rows[it];
        const int64_t lo = indptr[u];
        const int d = (int)(indptr[u + 1] - lo);
        __syncthreads();
        for (int e = tid; e < f * LDB; e += 256) B[e] = (e / LDB == e % LDB) ? 1.f : 0.f;
        for (int base = 0; base < d; base += RC) {
            const int nrow = min(RC, d - base);
            __syncthreads();
            for (int e = tid; e < nrow * nch; e += 256) {
                const int j = e / nch, c = e % nch;
                const int idx = indices[lo + base + j];
                *reinterpret_cast<float4*>(&Vs[j * LDS_LD + 4 * c]) = reinterpret_cast<const float4*>(V + (int64_t)idx * ld)[c];
            }
            if (tid < nrow) {
                const int idx = indices[lo + base + tid];
                float wj = vals[lo + base + tid];
                if (biasv) wj -= biasv[idx];
                wsm[tid] = wj;
            }
            __syncthreads();
            for (int e = tid; e < f * LDB; e += 256) {
                const int a = e / LDB, b = e % LDB;
                float s = 0.f;
                if (b < f) { for (int j = 0; j < nrow; ++j) s += wsm[j] * Vs[j * LDS_LD + a] * Vs[j * LDS_LD + b]; }
                else       { for (int j = 0; j < nrow; ++j) s += (wsm[j] + 1.f) * Vs[j * LDS_LD + a]; }
                B[e] += s;
            }
        }
        __syncthreads();
        bool singular = false;
        for (int k = 0; k < f; ++k) {
            float best = -1.f; int bi = k;
            for (int i = k + tid; i < f; i += 256) {
                const float v = fabsf(B[i * LDB + k]);
                if (v > best) { best = v; bi = i; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if ((tid & 63) == 0) { red[(tid >> 6) * 2] = best; red[(tid >> 6) * 2 + 1] = __int_as_float(bi); }
            __syncthreads();
            best = red[0]; bi = __float_as_int(red[1]);
#pragma unroll
            for (int wv = 1; wv < 4; ++wv) {
                const float ob = red[wv * 2]; const int oi = __float_as_int(red[wv * 2 + 1]);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            __syncthreads();                                    // red[] is rewritten in the next step
            if (!(best > 1e-30f)) { singular = true; break; }   // uniform
            if (bi != k) {
                for (int c = tid; c < LDB; c += 256) { const float a = B[k * LDB + c]; B[k * LDB + c] = B[bi * LDB + c]; B[bi * LDB + c] = a; }
                __syncthreads();
            }
            const float inv = 1.f / B[k * LDB + k];
            const int ty = tid >> 4, tx = tid & 15;
            // every row's multiplier is read before any thread overwrites column k: column k itself is not updated
            for (int i = k + 1 + ty; i < f; i += 16) {
                const float l = B[i * LDB + k] * inv;
                for (int c = k + 1 + tx; c < LDB; c += 16) B[i * LDB + c] -= l * B[k * LDB + c];
            }
            __syncthreads();
        }
        if (singular) {
            if (tid == 0) atomicAdd(fail_count, 1);
            for (int c = tid; c < ld; c += 256) g[(int64_t)u * ld + c] = 0.f;
            continue;
        }
        for (int k = f - 1; k >= 0; --k) {                      // back substitution (column oriented)
            const float xk = B[k * LDB + f] / B[k * LDB + k];
            __syncthreads();
            for (int i = tid; i < k; i += 256) B[i * LDB + f] -= B[i * LDB + k] * xk;
            if (tid == 0) B[k * LDB + f] = xk;
            __syncthreads();
        }
        for (int c = tid; c < ld; c += 256) g[(int64_t)u * ld + c] = (c < f) ? B[c * LDB + f] : 0.f;
    }
}

template <int NFB>
static void launch_wide_nfb(const int32_t* rows, int64_t count, const float* V, const float* biasv, const int64_t* indptr,
                            const int32_t* indices, const float* vals, int f, int ld, float* g, int32_t* fb_rows,
                            int32_t* fb_count, hipStream_t st, const int32_t* count_dev) {
    using C = WideCfg<NFB>;
    constexpr size_t lds = (size_t)C::TOTAL * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)solve_wide_kernel<NFB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    int64_t grid = 256 * 2;                                      // one resident workgroup per CU (LDS), two rounds
    if (grid > count) grid = count;
    static const char* nm = wmf_kname("solve_wide_kernel<%d>", NFB);
    static const char* nmb = wmf_kname("solve_wide_kernel<%d> [bounced]", NFB);
    WMF_LAUNCH(count_dev ? nmb : nm, (solve_wide_kernel<NFB>), dim3((unsigned)grid), dim3(C::NTHR), lds, st, rows, count, V, biasv, indptr,
               indices, vals, f, ld, g, fb_rows, fb_count, wmf_debug_flags, count_dev);
}

int wmf_wide_supported(int f) { return f > 144 && f <= 272; }

int wmf_launch_wide(const int32_t* rows, int64_t count, const float* V, const float* biasv, const int64_t* indptr,
                    const int32_t* indices, const float* vals, int f, int ld, float* g, int32_t* fb_rows,
                    int32_t* fb_count, hipStream_t st, const int32_t* count_dev) {
    if (count <= 0) return 0;
    switch ((f + 15) / 16) {
#define C_(N) case N: launch_wide_nfb<N>(rows, count, V, biasv, indptr, indices, vals, f, ld, g, fb_rows, fb_count, st, count_dev); break;
        C_(10) C_(11) C_(12) C_(13) C_(14) C_(15) C_(16) C_(17)
#undef C_
        default: return -1;
    }
    return 0;
}

size_t wmf_wide_lu_workspace_bytes(int f) { return (size_t)WMF_WIDE_LU_GRID * f * (f + 1) * sizeof(float); }

int wmf_launch_wide_lu(const int32_t* rows, const int32_t* count_ptr, const float* V, const float* biasv,
                       const int64_t* indptr, const int32_t* indices, const float* vals, int f, int ld, float* g,
                       int32_t* fail_count, float* work, hipStream_t st) {
    if (f > 272 || ld > 276) return -1;
    WMF_LAUNCH("solve_wide_lu_kernel", solve_wide_lu_kernel, dim3(WMF_WIDE_LU_GRID), dim3(256), 0, st, rows, count_ptr, V, biasv,
               indptr, indices, vals, f, ld, g, fail_count, work);
    return 0;
}
