// Rows with many stored entries when the factor width needs more than one wave (64 < f <= 144):
// the f x f whitened system  (I + V_u^T D V_u) g = V_u^T p  is built by f32 MFMA and solved by a blocked
// (16 x 16) Cholesky factorisation whose tiles never leave the MFMA accumulator registers (gfx950).
//
// Reference arithmetic: RecModel/wmf_model.py:233-239 -- Y_rel^T (Y_rel * data) accumulated per row and
// np.linalg.solve of the k x k system; the system is symmetric positive definite whenever the weights
// are non-negative, so Cholesky replaces LU.  A row whose system is not positive definite (possible
// with biases) is bounced to the pivoted LU kernel.
//
// One 256-thread workgroup per row.  B = R^T R with R upper triangular; tile (bi, bj), bi <= bj, is a
// 16 x 16 MFMA accumulator: lane (r = l & 15, q = l >> 4) holds B[16 bi + 4q + reg][16 bj + r].  The right
// hand side rides along as one more block column (bj = NFB), so the forward substitution is part of
// the factorisation.  Wave w owns the tiles t = w (mod 4).
//   A. entries are staged 32 at a time through LDS (two register sets keep two chunks in flight, the
//      next row's first chunks are requested before this row's factorisation);
//      tile(bi, bj) += frag[bi]^T (w * frag[bj]),  tile(bi, NFB) += frag[bi]^T (p in column 0).
//   C. for each block row p:  the owner of tile (p,p) publishes it, wave 0 factors it (one lane per row,
//      v_readlane broadcasts) and inverts the factor;  R_pj = L_pp^-1 B_pj is FOUR MFMAs per tile with
//      the tile's own accumulator registers as the B operand (k = 4q + reg is exactly its layout);
//      the row panel goes to a 16 x 16 (NFB+1) LDS buffer;  B_ij -= R_pi^T R_pj reads both operands
//      from that buffer and accumulates in place.  LDS holds one panel, never the matrix.
//   D. backward substitution R g = y with the stored inverse diagonal blocks; tile contributions are
//      reduced by DPP row sums and LDS float atomics.
#include "wmf_common.h"
#include "wmf_internal.h"
#include "wmf_tile.h"

template <int NFB>
struct DirectCfg {
    static constexpr int FP = 16 * NFB;
    static constexpr int LDV = (FP % 32 == 16) ? FP : FP + 16;   // = 16 (mod 32): the two k rows of a half wave hit disjoint banks
    static constexpr int RC = 32;
    static constexpr int NT = NFB * (NFB + 1) / 2 + NFB;          // upper tiles + one rhs tile per block row
    static constexpr int NACC = (NT + 3) / 4;
    static constexpr int PF = (RC * (FP / 4) + 255) / 256;        // 16-byte pieces prefetched per thread
    // LDS carve (floats)
    static constexpr int OFF_VS = 0;
    static constexpr int OFF_W = OFF_VS + RC * LDV;
    static constexpr int OFF_P = OFF_W + RC;
    static constexpr int OFF_D = OFF_P + RC;                       // [16][20] diagonal tile being factored
    static constexpr int OFF_T = OFF_D + 320;                      // [NFB][16][20] inverse diagonal factors
    static constexpr int OFF_PAN = OFF_T + NFB * 320;              // [NFB + 1][16][20] row panel
    static constexpr int OFF_Z = OFF_PAN + (NFB + 1) * 320;        // [FP] running rhs of the backward substitution
    static constexpr int OFF_G = OFF_Z + FP;                       // [FP] solution
    static constexpr int OFF_FLAG = OFF_G + FP;                    // [4]
    static constexpr int TOTAL = OFF_FLAG + 4;
};

// tile index of (bi, bj), bi <= bj <= NFB, in the enumeration "for bi: for bj = bi .. NFB"
template <int NFB>
__device__ __host__ constexpr int tile_index(int bi, int bj) {
    return bi * (NFB + 1) - (bi * (bi - 1)) / 2 + (bj - bi);
}

// MFMA accumulation of one staged chunk for wave S of 4
template <int NFB, int S>
__device__ __forceinline__ void direct_chunk(const float* __restrict__ Vs, const float* __restrict__ wsm,
                                             const float* __restrict__ psm, int nsteps,
                                             f32x4 (&acc)[DirectCfg<NFB>::NACC], int r, int q) {
    using C = DirectCfg<NFB>;
    for (int s = 0; s < nsteps; ++s) {
        const float wq = wsm[4 * s + q];
        const float pb = (r == 0) ? psm[4 * s + q] : 0.f;          // rhs tile: p in column 0
        const float* vrow = Vs + (4 * s + q) * C::LDV + r;
        float fr[NFB], fw[NFB];
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) { fr[fb] = vrow[16 * fb]; fw[fb] = fr[fb] * wq; }
        int t = 0;
#pragma unroll
        for (int bi = 0; bi < NFB; ++bi) {
#pragma unroll
            for (int bj = bi; bj <= NFB; ++bj, ++t) {
                if (t % 4 == S) acc[t / 4] = WMF_MFMA16(fr[bi], bj < NFB ? fw[bj] : pb, acc[t / 4]);
            }
        }
    }
}

// Phase C + D for wave S (compile time so that accumulator indices are static)
template <int NFB, int S>
__device__ __forceinline__ void direct_factor(float* __restrict__ sm, f32x4 (&acc)[DirectCfg<NFB>::NACC], int lane, int r, int q) {
    using C = DirectCfg<NFB>;
    float* Dblk = sm + C::OFF_D; float* T = sm + C::OFF_T; float* Pan = sm + C::OFF_PAN;
    float* zb = sm + C::OFF_Z; float* gs = sm + C::OFF_G; int* flag = reinterpret_cast<int*>(sm + C::OFF_FLAG);
    // identity on the diagonal tiles
#pragma unroll
    for (int b = 0; b < NFB; ++b) {
        const int t = tile_index<NFB>(b, b);
        if (t % 4 == S) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) if (r == 4 * q + reg) acc[t / 4][reg] += 1.f;
        }
    }
#pragma unroll
    for (int p = 0; p < NFB; ++p) {
        // (a) publish the diagonal tile
        {
            const int t = tile_index<NFB>(p, p);
            if (t % 4 == S) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) Dblk[(4 * q + reg) * 20 + r] = acc[t / 4][reg];
            }
        }
        __syncthreads();
        // (b) wave 0 factors it
        if (S == 0) {
            const bool ok = direct_diag(Dblk, T + p * 320, lane);
            if (!ok && lane == 0) flag[0] = 1;
        }
        __syncthreads();
        // (c) row panel: R_pj = X B_pj, the tile itself is the B operand
        {
            const float4 x4 = *reinterpret_cast<const float4*>(T + p * 320 + r * 20 + 4 * q);
#pragma unroll
            for (int j = p + 1; j <= NFB; ++j) {
                const int t = tile_index<NFB>(p, j);
                if (t % 4 == S) {
                    f32x4 n = f32x4{0.f, 0.f, 0.f, 0.f};
                    n = WMF_MFMA16(x4.x, acc[t / 4][0], n); n = WMF_MFMA16(x4.y, acc[t / 4][1], n);
                    n = WMF_MFMA16(x4.z, acc[t / 4][2], n); n = WMF_MFMA16(x4.w, acc[t / 4][3], n);
                    acc[t / 4] = n;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) Pan[j * 320 + (4 * q + reg) * 20 + r] = n[reg];
                }
            }
        }
        __syncthreads();
        // (e) trailing update  B_ij -= R_pi^T R_pj,  p < i <= j <= NFB, i < NFB
#pragma unroll
        for (int i = p + 1; i < NFB; ++i) {
#pragma unroll
            for (int j = i; j <= NFB; ++j) {
                const int t = tile_index<NFB>(i, j);
                if (t % 4 == S) {
                    f32x4 c = acc[t / 4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = Pan[i * 320 + (4 * q + e) * 20 + r];
                        const float b = Pan[j * 320 + (4 * q + e) * 20 + r];
                        c = WMF_MFMA16(-a, b, c);
                    }
                    acc[t / 4] = c;
                }
            }
        }
        __syncthreads();
    }
    // y_p = column 0 of tile (p, NFB)
#pragma unroll
    for (int p = 0; p < NFB; ++p) {
        const int t = tile_index<NFB>(p, NFB);
        if (t % 4 == S && r == 0) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) zb[16 * p + 4 * q + reg] = acc[t / 4][reg];
        }
    }
    __syncthreads();
    // ---- D: backward substitution  R g = y
#pragma unroll
    for (int p = NFB - 1; p >= 0; --p) {
        if (S == 0) {                                            // g_p = X_p^T z_p : lane (c' = r, q) sums c = q, q+4, ...
            float s = 0.f;
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) { const int c = q + 4 * cc; s += T[p * 320 + c * 20 + r] * zb[16 * p + c]; }
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (q == 0) gs[16 * p + r] = s;
        }
        __syncthreads();
        // z_i -= R_ip g_p for the tiles (i, p), i < p, in column p
        const float gp = gs[16 * p + r];
#pragma unroll
        for (int i = 0; i < p; ++i) {
            const int t = tile_index<NFB>(i, p);
            if (t % 4 == S) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    const float v = wmf_row16_sum(acc[t / 4][reg] * gp);       // sum over the 16 columns of the tile
                    if (r == 0) atomicAdd(&zb[16 * i + 4 * q + reg], -v);
                }
            }
        }
        __syncthreads();
    }
}

template <int NFB>
__global__ __launch_bounds__(256, (NFB <= 4 ? 3 : 2)) void solve_direct_kernel(const int32_t* __restrict__ rows, int64_t count,
                                                           const float* __restrict__ V, const float* __restrict__ biasv,
                                                           const int64_t* __restrict__ indptr,
                                                           const int32_t* __restrict__ indices,
                                                           const float* __restrict__ vals, int f, int ld,
                                                           float* __restrict__ g, int32_t* __restrict__ fb_rows,
                                                           int32_t* __restrict__ fb_count, int dbg) {
    using C = DirectCfg<NFB>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sm = reinterpret_cast<float*>(smem_raw);
    float* Vs = sm + C::OFF_VS; float* wsm = sm + C::OFF_W; float* psm = sm + C::OFF_P;
    float* gs = sm + C::OFF_G; int* flag = reinterpret_cast<int*>(sm + C::OFF_FLAG);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int nch = ld >> 2;

    // per-thread (row-in-chunk, piece) of each prefetched 16-byte piece; fixed across chunks
    int pj[C::PF], pc[C::PF];
#pragma unroll
    for (int i = 0; i < C::PF; ++i) { const int e = tid + 256 * i; pj[i] = e / nch; pc[i] = e % nch; }

    // Two register sets keep two staged chunks in flight; the first two chunks of the NEXT row are
    // requested before this row's factorisation starts, so the gather latency hides behind it.
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
    for (int e = tid; e < C::RC * C::LDV; e += 256) Vs[e] = 0.f;       // pad columns [ld, LDV) stay zero for good

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
        for (int i = 0; i < C::NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (tid == 0) flag[0] = 0;

        // slot 0 = set A (even chunks), slot 1 = set B (odd chunks); the next row's chunk `slot` goes
        // into the same set once this row no longer needs it
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
            const int nsteps = (dbg & 2) ? 0 : (nrow + 3) >> 2;
            switch (wave) {
                case 0: direct_chunk<NFB, 0>(Vs, wsm, psm, nsteps, acc, r, q); break;
                case 1: direct_chunk<NFB, 1>(Vs, wsm, psm, nsteps, acc, r, q); break;
                case 2: direct_chunk<NFB, 2>(Vs, wsm, psm, nsteps, acc, r, q); break;
                default: direct_chunk<NFB, 3>(Vs, wsm, psm, nsteps, acc, r, q); break;
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
        __syncthreads();
        if (!(dbg & 1)) {
            switch (wave) {
                case 0: direct_factor<NFB, 0>(sm, acc, lane, r, q); break;
                case 1: direct_factor<NFB, 1>(sm, acc, lane, r, q); break;
                case 2: direct_factor<NFB, 2>(sm, acc, lane, r, q); break;
                default: direct_factor<NFB, 3>(sm, acc, lane, r, q); break;
            }
        }
        __syncthreads();
        const bool notpd = flag[0] != 0;
        if (notpd) {
            if (tid == 0) fb_rows[atomicAdd(fb_count, 1)] = u;   // not positive definite: the LU kernel redoes it
        } else {
            for (int c = tid; c < ld; c += 256) g[(int64_t)u * ld + c] = (c < f) ? gs[c] : 0.f;
        }
        u = un; lo = lon; d = dn;
        __syncthreads();                                         // gs / flag are reused by the next row
    }
}

template <int NFB>
static void launch_direct_nfb(const int32_t* rows, int64_t count, const float* V, const float* biasv,
                              const int64_t* indptr, const int32_t* indices, const float* vals, int f, int ld, float* g,
                              int32_t* fb_rows, int32_t* fb_count, int dbg, hipStream_t st) {
    using C = DirectCfg<NFB>;
    constexpr size_t lds = (size_t)C::TOTAL * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)solve_direct_kernel<NFB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int per_cu = (int)(160 * 1024 / lds) > 0 ? (int)(160 * 1024 / lds) : 1;
    int64_t grid = 256 * (int64_t)(per_cu > 4 ? 4 : per_cu) * 2;
    if (grid > count) grid = count;
    hipLaunchKernelGGL((solve_direct_kernel<NFB>), dim3((unsigned)grid), dim3(256), lds, st, rows, count, V, biasv, indptr,
                       indices, vals, f, ld, g, fb_rows, fb_count, dbg);
}

int wmf_direct_supported(int f) { return f >= 1 && f <= 144; }

int wmf_launch_direct(const int32_t* rows, int64_t count, const float* V, const float* biasv, const int64_t* indptr,
                      const int32_t* indices, const float* vals, int f, int ld, float* g, int32_t* fb_rows,
                      int32_t* fb_count, hipStream_t st) {
    if (count <= 0) return 0;
    const int dbg = wmf_debug_flags;
    switch ((f + 15) / 16) {
#define C_(N) case N: launch_direct_nfb<N>(rows, count, V, biasv, indptr, indices, vals, f, ld, g, fb_rows, fb_count, dbg, st); break;
        C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9)
#undef C_
        default: return -1;
    }
    return 0;
}
