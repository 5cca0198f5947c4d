// Rows with many stored entries: the f x f whitened system  (I + V_u^T D V_u) g = V_u^T p  built by
// f32 MFMA and solved by a blocked (16 x 16) Cholesky factorisation that also runs on MFMA (gfx950).
//
// Reference arithmetic: RecModel/wmf_model.py:233-239 -- Y_rel^T (Y_rel * data) accumulated per row
// and np.linalg.solve of the k x k system; here the system is symmetric positive definite whenever
// the weights are non-negative, so Cholesky replaces LU.  A row whose system is not positive
// definite (possible with biases) is bounced to the pivoted LU kernel.
//
// One 256-thread workgroup per row:
//   A. entries are staged RC at a time through LDS (prefetched one chunk ahead in registers);
//      wave w accumulates the upper tiles t = w (mod 4) of V^T D V:  tile(bi,bj) += frag[bi]^T (w*frag[bj]);
//      threads 0..FP-1 accumulate rhs = V^T p.
//   B. tiles (+ I) are written transposed into the lower triangle of Bm[FP+16][LDB]; rhs becomes an
//      extra block row, so the forward substitution is just part of the factorisation.
//   C. for each 16-column block: wave 0 factors the diagonal block and inverts its factor in
//      registers (one lane per row / per column); all waves do the panel (A_ik Linv_kk^T) and the
//      trailing update (A_ij -= L_ik L_jk^T) with 4 MFMAs per 16 x 16 tile.
//   D. wave 0 back-substitutes with the stored Linv_kk blocks; the row of g is written.
#include "wmf_common.h"
#include "wmf_internal.h"

__device__ __forceinline__ float dreadlane(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

template <int NFB>
struct DirectCfg {
    static constexpr int FP = 16 * NFB;
    static constexpr int LDV = (FP % 32 == 16) ? FP : FP + 16;   // = 16 (mod 32): the two k rows of a half wave hit disjoint banks
    static constexpr int LDB = FP + 4;
    static constexpr int RC = 32;
    static constexpr int NT = NFB * (NFB + 1) / 2;
    static constexpr int NACC = (NT + 3) / 4;
    static constexpr int PF = (RC * (FP / 4) + 255) / 256;        // 16-byte pieces prefetched per thread
    // LDS carve (floats)
    static constexpr int OFF_VS = 0;
    static constexpr int OFF_W = OFF_VS + RC * LDV;
    static constexpr int OFF_P = OFF_W + RC;
    static constexpr int OFF_T = OFF_P + RC;                       // [NFB][16][20] inverse diagonal factors
    static constexpr int OFF_G = OFF_T + NFB * 320;                // [FP] solution
    static constexpr int OFF_FLAG = OFF_G + FP;                    // [4]
    static constexpr int OFF_B = OFF_FLAG + 4;                     // [FP + 16][LDB]
    static constexpr int TOTAL = OFF_B + (FP + 16) * LDB;
};

// MFMA accumulation of one staged chunk for wave S of 4
template <int NFB, int S>
__device__ __forceinline__ void direct_chunk(const float* __restrict__ Vs, const float* __restrict__ wsm, int nsteps,
                                             f32x4 (&acc)[DirectCfg<NFB>::NACC], int r, int q) {
    using C = DirectCfg<NFB>;
    for (int s = 0; s < nsteps; ++s) {
        const float wq = wsm[4 * s + q];
        const float* vrow = Vs + (4 * s + q) * C::LDV + r;
        float fr[NFB], fw[NFB];
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) { fr[fb] = vrow[16 * fb]; fw[fb] = fr[fb] * wq; }
        int t = 0;
#pragma unroll
        for (int bi = 0; bi < NFB; ++bi) {
#pragma unroll
            for (int bj = bi; bj < NFB; ++bj, ++t) {
                if (t % 4 == S) acc[t / 4] = WMF_MFMA16(fr[bi], fw[bj], acc[t / 4]);
            }
        }
    }
}

// write wave S's tiles (+ identity) transposed into the lower triangle of Bm
template <int NFB, int S>
__device__ __forceinline__ void direct_store(float* __restrict__ Bm, const f32x4 (&acc)[DirectCfg<NFB>::NACC], int r, int q) {
    using C = DirectCfg<NFB>;
    int t = 0;
#pragma unroll
    for (int bi = 0; bi < NFB; ++bi) {
#pragma unroll
        for (int bj = bi; bj < NFB; ++bj, ++t) {
            if (t % 4 == S) {
                float4 v = make_float4(acc[t / 4][0], acc[t / 4][1], acc[t / 4][2], acc[t / 4][3]);
                if (bi == bj) {
                    if (r == 4 * q + 0) v.x += 1.f;
                    if (r == 4 * q + 1) v.y += 1.f;
                    if (r == 4 * q + 2) v.z += 1.f;
                    if (r == 4 * q + 3) v.w += 1.f;
                }
                *reinterpret_cast<float4*>(&Bm[(16 * bj + r) * C::LDB + 16 * bi + 4 * q]) = v;
            }
        }
    }
}

// wave 0: Cholesky of the 16 x 16 diagonal block kb and the inverse of its factor.  Returns false
// (uniformly) if a pivot is not positive.
template <int LDB>
__device__ __forceinline__ bool direct_diag(float* __restrict__ Bm, float* __restrict__ T, int kb, int lane) {
    float a[16];
    const int row = (lane < 16) ? lane : 0;
    const float* src = Bm + (16 * kb + row) * LDB + 16 * kb;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(src + 4 * c);
        a[4 * c] = v.x; a[4 * c + 1] = v.y; a[4 * c + 2] = v.z; a[4 * c + 3] = v.w;
    }
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float dk = dreadlane(a[k], k);
        if (!(dk > 1e-20f)) ok = false;
        const float inv = __builtin_amdgcn_rsqf(dk), s = dk * inv;
        a[k] = (lane == k) ? s : a[k] * inv;
#pragma unroll
        for (int j = k + 1; j < 16; ++j) a[j] -= a[k] * dreadlane(a[k], j);
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) if (j > lane) a[j] = 0.f;
    // inverse: lane j builds column j of X = L^-1
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        float s = (i == lane) ? 1.f : 0.f;
#pragma unroll
        for (int k = 0; k < i; ++k) s -= dreadlane(a[k], i) * x[k];
        x[i] = s * __builtin_amdgcn_rcpf(dreadlane(a[i], i));
    }
    if (lane < 16) {
        float* dst = Bm + (16 * kb + lane) * LDB + 16 * kb;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            *reinterpret_cast<float4*>(dst + 4 * c) = make_float4(a[4 * c], a[4 * c + 1], a[4 * c + 2], a[4 * c + 3]);
        float* t = T + kb * 320;
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i * 20 + lane] = x[i];      // T[i][j] = Linv[i][j]
    }
    return ok;
}

template <int NFB>
__global__ __launch_bounds__(256, (NFB <= 4 ? 4 : (NFB <= 6 ? 2 : 1))) void solve_direct_kernel(const int32_t* __restrict__ rows, int64_t count,
                                                           const float* __restrict__ V, const float* __restrict__ biasv,
                                                           const int64_t* __restrict__ indptr,
                                                           const int32_t* __restrict__ indices,
                                                           const float* __restrict__ vals, int f, int ld,
                                                           float* __restrict__ g, int32_t* __restrict__ fb_rows,
                                                           int32_t* __restrict__ fb_count, int dbg) {
    using C = DirectCfg<NFB>;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sm = reinterpret_cast<float*>(smem_raw);
    float* Vs = sm + C::OFF_VS; float* wsm = sm + C::OFF_W; float* psm = sm + C::OFF_P; float* T = sm + C::OFF_T;
    float* gs = sm + C::OFF_G; int* flag = reinterpret_cast<int*>(sm + C::OFF_FLAG); float* Bm = sm + C::OFF_B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int nch = ld >> 2;
    const int NB = (f + 15) >> 4;                  // blocks that can be non-trivial (== NFB)

    // per-thread (row-in-chunk, piece) of each prefetched 16-byte piece; fixed across chunks
    int pj[C::PF], pc[C::PF];
#pragma unroll
    for (int i = 0; i < C::PF; ++i) { const int e = tid + 256 * i; pj[i] = e / nch; pc[i] = e % nch; }

    // Two register sets keep two staged chunks in flight; the first two chunks of the NEXT row are
    // requested before this row's factorisation starts, so the gather latency hides behind it.
    float4 preA[C::PF], preB[C::PF];
    float wA = 0.f, wB = 0.f;
    auto load_chunk = [&](float4 (&pre)[C::PF], float& wpre, int64_t lo, int d, int base) {
        const int nrow = min(C::RC, d - base);                   // may be <= 0: loads nothing
#pragma unroll
        for (int i = 0; i < C::PF; ++i) {                        // unconditional loads, masked afterwards
            const bool on = pj[i] < nrow;
            const int idx = indices[on ? lo + base + pj[i] : 0];
            const float4 v = reinterpret_cast<const float4*>(V + (int64_t)idx * ld)[pc[i]];
            pre[i] = on ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        {
            const bool on = tid < nrow;
            const int64_t e = on ? lo + base + tid : 0;
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
        // next row's metadata (its first chunks are requested once this row's last chunk is staged)
        const int64_t itn = it + gridDim.x;
        int un = 0, dn = 0;
        int64_t lon = 0;
        if (itn < count) { un = rows[itn]; lon = indptr[un]; dn = (int)(indptr[un + 1] - lon); }

        f32x4 acc[C::NACC];
#pragma unroll
        for (int i = 0; i < C::NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        float racc = 0.f;
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
                case 0: direct_chunk<NFB, 0>(Vs, wsm, nsteps, acc, r, q); break;
                case 1: direct_chunk<NFB, 1>(Vs, wsm, nsteps, acc, r, q); break;
                case 2: direct_chunk<NFB, 2>(Vs, wsm, nsteps, acc, r, q); break;
                default: direct_chunk<NFB, 3>(Vs, wsm, nsteps, acc, r, q); break;
            }
            if (tid < C::FP && !(dbg & 4)) {
                float s = 0.f;
                for (int j = 0; j < nrow; ++j) s += psm[j] * Vs[j * C::LDV + tid];
                racc += s;
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
        // ---- B: matrix (+ I) into the lower triangle, rhs as block row NFB (row FP), rest of it zero
        switch (wave) {
            case 0: direct_store<NFB, 0>(Bm, acc, r, q); break;
            case 1: direct_store<NFB, 1>(Bm, acc, r, q); break;
            case 2: direct_store<NFB, 2>(Bm, acc, r, q); break;
            default: direct_store<NFB, 3>(Bm, acc, r, q); break;
        }
        for (int e = tid; e < 16 * C::LDB; e += 256) {
            const int rr = e / C::LDB, cc = e % C::LDB;
            Bm[(C::FP + rr) * C::LDB + cc] = 0.f;
        }
        __syncthreads();
        if (tid < C::FP) Bm[C::FP * C::LDB + tid] = racc;
        __syncthreads();

        // ---- C: blocked Cholesky of the leading NB blocks; block row NFB (the rhs) rides along
        for (int kb = 0; kb < ((dbg & 1) ? 0 : NB); ++kb) {
            if (wave == 0 && !(dbg & 8)) {
                const bool ok = direct_diag<C::LDB>(Bm, T, kb, lane);
                if (!ok && lane == 0) flag[0] = 1;
            }
            __syncthreads();
            // panel: L_ik = A_ik Linv_kk^T for block rows ib in (kb, NB) and the rhs row (ib = NFB)
            {
                const float4 b4 = *reinterpret_cast<const float4*>(&T[kb * 320 + r * 20 + 4 * q]);
                int t = 0;
                for (int ib = kb + 1; ib <= NB; ++ib, ++t) {
                    if ((t & 3) != wave) continue;
                    const int brow = (ib == NB) ? NFB : ib;      // rhs block row sits at NFB
                    const float4 a4 = *reinterpret_cast<const float4*>(&Bm[(16 * brow + r) * C::LDB + 16 * kb + 4 * q]);
                    f32x4 p = f32x4{0.f, 0.f, 0.f, 0.f};
                    p = WMF_MFMA16(a4.x, b4.x, p); p = WMF_MFMA16(a4.y, b4.y, p);
                    p = WMF_MFMA16(a4.z, b4.z, p); p = WMF_MFMA16(a4.w, b4.w, p);
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) Bm[(16 * brow + 4 * q + reg) * C::LDB + 16 * kb + r] = p[reg];
                }
            }
            __syncthreads();
            // trailing update: A_ij -= L_ik L_jk^T, kb < jb <= ib, jb < NB
            {
                int t = 0;
                for (int ib = kb + 1; ib <= NB; ++ib) {
                    const int brow = (ib == NB) ? NFB : ib;
                    for (int jb = kb + 1; jb <= ib && jb < NB; ++jb, ++t) {
                        if ((t & 3) != wave) continue;
                        const float4 a4 = *reinterpret_cast<const float4*>(&Bm[(16 * brow + r) * C::LDB + 16 * kb + 4 * q]);
                        const float4 b4 = *reinterpret_cast<const float4*>(&Bm[(16 * jb + r) * C::LDB + 16 * kb + 4 * q]);
                        f32x4 p;
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) p[reg] = Bm[(16 * brow + 4 * q + reg) * C::LDB + 16 * jb + r];
                        p = WMF_MFMA16(-a4.x, b4.x, p); p = WMF_MFMA16(-a4.y, b4.y, p);
                        p = WMF_MFMA16(-a4.z, b4.z, p); p = WMF_MFMA16(-a4.w, b4.w, p);
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) Bm[(16 * brow + 4 * q + reg) * C::LDB + 16 * jb + r] = p[reg];
                    }
                }
            }
            __syncthreads();
        }
        const bool notpd = flag[0] != 0;
        if (notpd && tid == 0) fb_rows[atomicAdd(fb_count, 1)] = u;   // not positive definite: the LU kernel redoes it
        // ---- D: back substitution  g = L^-T y,  y = row FP of Bm.  Wave 0, lane (c = r, part = q).
        if (wave == 0 && !notpd && !(dbg & 1)) {
            for (int kb = NB - 1; kb >= 0; --kb) {
                float z = 0.f;
                for (int i = 16 * (kb + 1) + q; i < 16 * NB; i += 4) z += Bm[i * C::LDB + 16 * kb + r] * gs[i];
                z += __shfl_xor(z, 16);
                z += __shfl_xor(z, 32);
                z = Bm[C::FP * C::LDB + 16 * kb + r] - z;         // every lane (r, *) now holds z_r
                // g_c' = sum_c Linv[c][c'] z_c : lane (c' = r, q) sums c = q, q+4, ...
                float s = 0.f;
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    const int c = q + 4 * cc;
                    s += T[kb * 320 + c * 20 + r] * __shfl(z, c);
                }
                s += __shfl_xor(s, 16);
                s += __shfl_xor(s, 32);
                if (q == 0) gs[16 * kb + r] = s;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        __syncthreads();
        if (!notpd)
            for (int c = tid; c < ld; c += 256) g[(int64_t)u * ld + c] = (c < f) ? gs[c] : 0.f;
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
    int64_t grid = 256 * (int64_t)(per_cu > 8 ? 8 : per_cu) * 2;
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
