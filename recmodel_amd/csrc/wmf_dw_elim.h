// Block LDL^T elimination of the whitened f x f system held in MFMA accumulator tiles (phases C and D of the
// one-wave-per-row heavy-row kernels; see the header comment of wmf_directw.hip for the algebra and the layouts).
// In:  acc[tile_w(bi, bj)] = tile (bi, bj), bi <= bj, of V_u^T D V_u (the identity is added here); racc[fb] = this
//      lane's share (its q) of (V_u^T p)[16 fb + r]; BORDER: bacc / cacc / eacc = border column, corner, border rhs.
// Out: gb[p] = g[16 p + (lane & 15)] on every lane, tb = the border unknown, ok = false if a pivot was not positive.
// diag: what the identity is in the caller's units -- a caller whose tiles, right-hand side and border all carry a common
//      factor s (accumulated from operands scaled by sqrt(s)) passes diag = s; the solution is the same.
// Wv / Wb: two LDS vectors of NFB * 16 floats owned by this wave.  NOT __restrict__: lanes r == 0 write them under an exec
// mask and every lane reads them back; told that nothing else touches the memory, hipcc turns the masked store into
// load - select - store by ALL lanes, and the lanes that share an address then race with the one real writer.
#pragma once
#include "wmf_common.h"
#include "wmf_internal.h"

template <int NFB>
__device__ __host__ constexpr int tile_w(int bi, int bj) {       // "for bi: for bj = bi .. NFB - 1"
    return bi * NFB - (bi * (bi - 1)) / 2 + (bj - bi);
}

// partial systems of split rows / of the reduce-scatter exchange (layout: header comment of wmf_directw.hip)
#define WMF_DW_TRI 136
#define WMF_DW_TILES(NFB) (((NFB) * ((NFB) - 1) / 2) * 256 + (NFB) * WMF_DW_TRI)
#define WMF_DW_PARTIAL(NFB, BORDER) (WMF_DW_TILES(NFB) + ((NFB) + ((BORDER) ? (NFB) + 2 : 0)) * 64)
template <int NFB>
__device__ __host__ constexpr int tile_off(int bi, int bj) {     // float offset of tile (bi, bj) in a partial system
    const int t = bi * NFB - (bi * (bi - 1)) / 2 + (bj - bi), ndiag = bi + (bj > bi ? 1 : 0);
    return (t - ndiag) * 256 + ndiag * WMF_DW_TRI;
}

#ifndef WMF_DW_BP
#define WMF_DW_BP 0          // two-waves callers: multiplier column of the tile inverse by ds_bpermute instead of VALU lane swaps:
                             // five VALU instructions fewer a step, measured 19.49 against 19.26 ms at cfg3 (the permute's latency
                             // is not hidden when both waves of a SIMD sit in a sweep)
#endif
#ifndef WMF_DW_GJM
#define WMF_DW_GJM 0         // 1: two-waves callers invert tiles by the symmetric sweep with f32 MFMA rank-one updates (wmf_common.h).
                             // Measured 19.6 against 18.3 ms at cfg3: the f32 MFMA shares the VALU pipe (profiles/README.md), so its 32
                             // cycles cost more than the eight VALU instructions it replaces
#endif
#ifndef WMF_DW_OPAQUE
#define WMF_DW_OPAQUE 1
#endif

// WREG: w_p (and w^b_p) stay in registers instead of the two LDS vectors -- after the DPP row sums every lane (., q) holds
// w_p[4q + reg] already, which is all the backward pass reads (used by the LDS-DMA kernel, which keeps LDS accesses that the
// compiler can see out of the kernel).
// F16T: the tile products of the elimination (W_pj = X B_pj, B_ij -= B_pi^T W_pj: 448 f32 MFMAs at NFB = 8, 14 k cycles of
// the SIMD's one f32 pipe) as split-f16 MFMAs.  A 16 x 16 x 16 product uses half of a v_mfma_f32_16x16x32_f16's K = 32,
// so the other half carries the low parts: every tile is split once into f16 high and low parts, h + l (22 significand
// bits, 2^-23 relative), a lane's four values k = 4q + e give the operand { h_0..h_3, l_0..l_3 } (natural), and
//     A . B = mfma({ah, ah}, {bh, bl}) + mfma({al, al}, {bh, bl}) = (ah + al)(bh + bl)
// with no cross-lane movement at all (slot (q, jj) means k = 4q + (jj & 3) on every lane).  Two 16-cycle MFMAs that leave
// half their cycles to the VALU replace four 32-cycle ones; the splits cost 12 VALU instructions per tile of the pivot row.
typedef _Float16 dw_f16x8 __attribute__((ext_vector_type(8)));
// (Plain C++, not wmf_split4: these tiles are MFMA results, and hipcc keeps the wait states between an MFMA and a reader of its
// result only for instructions it can see into -- an inline-asm reader scheduled right behind the MFMA reads stale registers.)
__device__ __forceinline__ dw_f16x8 dw_split_natural(const f32x4 v) {
    dw_f16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma clang fp contract(off)
        const _Float16 h = (_Float16)v[e];
        o[e] = h;
        o[4 + e] = (_Float16)(v[e] - (float)h);
    }
    return o;
}
// for callers that run two waves per SIMD (RELANE): the asm split behind explicit wait states
__device__ __forceinline__ dw_f16x8 dw_split_natural_w(const f32x4 v) {
    return __builtin_bit_cast(dw_f16x8, wmf_split4_after_mfma(v[0], v[1], v[2], v[3]));
}
// The same product from the two natural operands themselves, as three K = 16 MFMAs on their halves (hi.hi, hi.lo, lo.hi; lo.lo
// is below 2^-22 of the product): no duplicated operands to build -- for the callers that are VALU bound with MFMA time to
// spare (two waves per SIMD).
typedef _Float16 dw_f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 dw_prod16(const dw_f16x8 a, const dw_f16x8 b, f32x4 c) {
    const dw_f16x4 ah = __builtin_shufflevector(a, a, 0, 1, 2, 3), al = __builtin_shufflevector(a, a, 4, 5, 6, 7);
    const dw_f16x4 bh = __builtin_shufflevector(b, b, 0, 1, 2, 3), bl = __builtin_shufflevector(b, b, 4, 5, 6, 7);
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, c, 0, 0, 0);
    return c;
}
#ifndef WMF_DW_K16
#define WMF_DW_K16 1
#endif
__device__ __forceinline__ dw_f16x8 dw_dup_hi(const dw_f16x8 n) { return __builtin_shufflevector(n, n, 0, 1, 2, 3, 0, 1, 2, 3); }
__device__ __forceinline__ dw_f16x8 dw_dup_lo(const dw_f16x8 n) { return __builtin_shufflevector(n, n, 4, 5, 6, 7, 4, 5, 6, 7); }

// RELANE: the lane coordinates (r, q) are formed again at every pivot from v_mbcnt in a volatile asm -- for the caller that
// runs two waves per SIMD on 256 registers: kept live through the elimination they are spilled, and every scratch reload is
// a VMEM operation that returns in order, behind all LDS-DMAs of the caller's ring.
template <int NFB, bool BORDER, bool GJ_LDS, bool WREG = false, bool F16T = false, bool RELANE = false>
__device__ __forceinline__ void dw_eliminate(f32x4 (&acc)[NFB * (NFB + 1) / 2], float (&racc)[NFB], float (&bacc)[BORDER ? NFB : 1],
                                             float& cacc, float& eacc, float* Wv, float* Wb, int r_in,
                                             int q_in, const int (&baddr)[4], int dbg, float (&gb)[NFB], float& tb, bool& ok,
                                             const float diag = 1.f) {
    int r = r_in, q = q_in;
    auto relane = [&]() {
        if constexpr (RELANE) {
            int l;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
            r = l & 15; q = l >> 4;
        }
    };
    relane();
    float wvs[WREG ? NFB : 1][4], wbs[(WREG && BORDER) ? NFB : 1][4];
    // RELANE callers keep LDS-DMAs in flight across the elimination: their w_p vectors go to LDS by inline asm (hipcc puts an
    // s_waitcnt vmcnt(0) in front of every LDS access it can see while an LDS-DMA may be outstanding)
    constexpr bool WASM = RELANE && !WREG;
    const unsigned wv_lds = WASM ? (unsigned)(uintptr_t)((__attribute__((address_space(3))) float*)Wv) : 0u;
    const unsigned wb_lds = (WASM && BORDER) ? (unsigned)(uintptr_t)((__attribute__((address_space(3))) float*)Wb) : 0u;
    auto lds_put4 = [&](unsigned base, int p, float a, float b, float c, float d) {     // every lane of group q: the same 16 bytes
        const f32x4 v = f32x4{a, b, c, d};
        asm volatile("ds_write_b128 %0, %1" ::"v"(base + (16 * p + 4 * q) * 4), "v"(v) : "memory");
    };
    auto lds_get4 = [&](unsigned base, int p) {
        f32x4 v;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(base + (16 * p + 4 * q) * 4) : "memory");
        return v;
    };
    int pmin = 0x7f800000, pmax = 0, spread = 0;                 // wave-uniform: smallest / largest pivot of the tile inverses (bit patterns), largest spread within a tile
    float cacc2 = 0.f, eacc2 = 0.f;                              // BORDER: per-lane parts of the two border sums
    // ---- C: block elimination, everything in registers except the two panel buffers
    if (!WMF_ABL(dbg, 1)) {
#pragma unroll
        for (int b = 0; b < NFB; ++b) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) if (r == 4 * q + reg) acc[tile_w<NFB>(b, b)][reg] += diag;
        }
        // LOOK-AHEAD: the inverse of pivot tile p + 1 is started as soon as row p + 1 of the trailing update has been
        // applied, in front of the remaining rows' MFMAs, which it does not depend on: with one wave per SIMD nothing else
        // hides the 16-step dependency chain of the Gauss-Jordan sweep.
        auto invert = [&](int p) {
            f32x4 X = acc[tile_w<NFB>(p, p)];
            // The sweep's pivot-row update  row += (1 / piv - 1) row  is exact to f32 only for pivots of order one (the
            // multiplier is rounded at 2^-24 of ONE, not of 1 / piv): a tile that carries the caller's factor `diag` is
            // brought back to unit scale for the sweep, and its inverse gets the factor once more -- it multiplies tiles
            // and vectors that still carry `diag`.
            if (diag != 1.f) X *= 1.f / diag;
            if constexpr (!GJ_LDS) {                                 // (the kernels whose wave runs alone on its SIMD)
                if constexpr (RELANE && WMF_DW_GJM != 0) {
                    gj_sweep_mfma(X, pmin, std::make_integer_sequence<int, 16>{});
                    X = -X;
                } else if (!WMF_ABL(dbg, 8)) {
                    float dsc = 1.f;                                 // un-normalised sweep (wmf_common.h): X = diag(dsc) . tile
                    gj_inv_sweep_lean<(RELANE && WMF_DW_BP != 0)>(X, dsc, pmin, pmax, spread, 4 * r, std::make_integer_sequence<int, 16>{});
                    X *= dsc;
                }
                if (diag != 1.f) X *= 1.f / diag;
                return X;
            }
#if WMF_DW_OPAQUE
            // the sweep's lane masks (r == K, q == K / 4) are the same for every pivot; hipcc hoists all 36 of them out of
            // the pivot loop and then spills them to VGPR lanes (v_writelane / v_readlane pairs around every use).  Lane
            // ids the compiler cannot see through make it compare in place: 20 v_cmp per pivot instead.
            int rp = r, qp = q;
            asm volatile("" : "+v"(rp), "+v"(qp));
            if (!WMF_ABL(dbg, 8))                                      // timing experiments only: 8 = no tile inverse
                gj_inv_sweep<GJ_LDS, true, true>(X, baddr, rp, qp, ok, std::make_integer_sequence<int, 16>{});
#else
            gj_inv_sweep<GJ_LDS, true, true>(X, baddr, r, q, ok, std::make_integer_sequence<int, 16>{});
#endif
            if (diag != 1.f) X *= 1.f / diag;
            return X;
        };
        f32x4 Xnext = RELANE ? f32x4{0.f, 0.f, 0.f, 0.f} : invert(0);       // (RELANE: no look-ahead, two waves hide the sweep's chain)
#pragma unroll
        for (int p = 0; p < NFB; ++p) {
            relane();
            const f32x4 X = RELANE ? invert(p) : Xnext;
            // y_p[r] complete (its four q shares added), then w_p = X y_p: lane (r, q) has X[4q + reg][r] (X is
            // symmetric), so the products summed over the 16 lanes of a DPP row give w_p[4q + reg] on the whole row
            float yp = racc[p];
            yp = wmf_qsum(yp);
            float wv0 = X[0] * yp, wv1 = X[1] * yp, wv2 = X[2] * yp, wv3 = X[3] * yp;
            wmf_row16_sum4(wv0, wv1, wv2, wv3);
            if constexpr (WREG) { wvs[p][0] = wv0; wvs[p][1] = wv1; wvs[p][2] = wv2; wvs[p][3] = wv3; }
            else if constexpr (WASM) lds_put4(wv_lds, p, wv0, wv1, wv2, wv3);
            else if (r == 0) *reinterpret_cast<float4*>(&Wv[16 * p + 4 * q]) = make_float4(wv0, wv1, wv2, wv3);
            float wb0 = 0.f, wb1 = 0.f, wb2 = 0.f, wb3 = 0.f;
            if constexpr (BORDER) {
                float bp = bacc[p];
                bp = wmf_qsum(bp);
                wb0 = X[0] * bp; wb1 = X[1] * bp; wb2 = X[2] * bp; wb3 = X[3] * bp;
                wmf_row16_sum4(wb0, wb1, wb2, wb3);
                if constexpr (WREG) { wbs[p][0] = wb0; wbs[p][1] = wb1; wbs[p][2] = wb2; wbs[p][3] = wb3; }
                else if constexpr (WASM) lds_put4(wb_lds, p, wb0, wb1, wb2, wb3);
                else if (r == 0) *reinterpret_cast<float4*>(&Wb[16 * p + 4 * q]) = make_float4(wb0, wb1, wb2, wb3);
                // b_p^T w^b_p and b_p^T w^y_p without moving anything: b_p[row] sits in lane r = row of every group and
                // w_p[4q + reg] on every lane of group q, so the lanes with r >> 2 == q hold one product each (reg = r & 3);
                // summed over the wave at the end
                {
                    const int sel = r & 3;
                    const float wbs_ = sel == 0 ? wb0 : sel == 1 ? wb1 : sel == 2 ? wb2 : wb3;
                    const float wvs_ = sel == 0 ? wv0 : sel == 1 ? wv1 : sel == 2 ? wv2 : wv3;
                    const float bm = (r >> 2) == q ? bp : 0.f;
                    cacc2 -= bm * wbs_;
                    eacc2 -= bm * wvs_;
                }
            }
            if constexpr (F16T) {
                // Row p in split f16 (header comment): W'_pj = -X B_pj replaces the tile (the sign makes the trailing update
                // an accumulation; the backward pass subtracts instead), y_j and b_j get their updates from the f32 tile
                // before it is overwritten.
                f32x4 nX;
#pragma unroll
                for (int e = 0; e < 4; ++e) nX[e] = -X[e];
                const dw_f16x8 xn = (RELANE ? dw_split_natural_w(nX) : dw_split_natural(nX)), xh = dw_dup_hi(xn), xl = dw_dup_lo(xn);
                dw_f16x8 nb[NFB], nw[NFB];
#pragma unroll
                for (int j = p + 1; j < NFB; ++j) {
                    const int t = tile_w<NFB>(p, j);
                    const f32x4 B = acc[t];
                    racc[j] -= B[0] * wv0 + B[1] * wv1 + B[2] * wv2 + B[3] * wv3;        // y_j[r] -= sum_rows B_pj[row][r] w_p[row]
                    if constexpr (BORDER) bacc[j] -= B[0] * wb0 + B[1] * wb1 + B[2] * wb2 + B[3] * wb3;
                    nb[j] = RELANE ? dw_split_natural_w(B) : dw_split_natural(B);
                    f32x4 n = f32x4{0.f, 0.f, 0.f, 0.f};
                    if constexpr (RELANE && WMF_DW_K16 != 0) n = dw_prod16(xn, nb[j], n);
                    else {
                        n = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, nb[j], n, 0, 0, 0);
                        n = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, nb[j], n, 0, 0, 0);
                    }
                    acc[t] = n;                                  // W'_pj = -W_pj stays in registers for the backward pass too
                    nw[j] = RELANE ? dw_split_natural_w(n) : dw_split_natural(n);
                }
#pragma unroll
                for (int i = p + 1; i < NFB; ++i) {
                    const dw_f16x8 ah = dw_dup_hi(nb[i]), al = dw_dup_lo(nb[i]);
#pragma unroll
                    for (int j = i; j < NFB; ++j) {
                        const int t = tile_w<NFB>(i, j);
                        f32x4 c = acc[t];
                        if constexpr (RELANE && WMF_DW_K16 != 0) c = dw_prod16(nb[i], nw[j], c);
                        else {
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, nw[j], c, 0, 0, 0);     // B_ij += B_pi^T W'_pj
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, nw[j], c, 0, 0, 0);
                        }
                        acc[t] = c;
                    }
                    if (!RELANE && i == p + 1) Xnext = invert(p + 1); // tile (p + 1, p + 1) is final: look ahead
                }
            } else {
                // Row p: W_pj = X B_pj replaces the tile, the original goes to `orig` for the trailing update.  Both operands
                // of that update are elements THIS lane already holds: instruction e of  B_ij -= B_pi^T W_pj  wants
                // A[m = r][k = q] = B_pi[4q + e][r] and B[k = q][n = r] = W_pj[4q + e][r], i.e. register e of the two tiles in
                // accumulator layout -- no LDS panel, no exchange.
                f32x4 orig[NFB];
    #pragma unroll
                for (int j = p + 1; j < NFB; ++j) {
                    const int t = tile_w<NFB>(p, j);
                    orig[j] = acc[t];
                    f32x4 n = f32x4{0.f, 0.f, 0.f, 0.f};
                    n = WMF_MFMA16(X[0], acc[t][0], n); n = WMF_MFMA16(X[1], acc[t][1], n);
                    n = WMF_MFMA16(X[2], acc[t][2], n); n = WMF_MFMA16(X[3], acc[t][3], n);
                    acc[t] = n;                                  // W_pj stays in registers for the backward pass too
                }
    #pragma unroll
                for (int i = p + 1; i < NFB; ++i) {
                    float a[4];
    #pragma unroll
                    for (int e = 0; e < 4; ++e) a[e] = -orig[i][e];
                    // y_i[r] -= sum_rows B_pi[row][r] w_p[row]: this lane's rows are 4q + e, a[e] = -B_pi[4q + e][r]
                    racc[i] += a[0] * wv0 + a[1] * wv1 + a[2] * wv2 + a[3] * wv3;
                    if constexpr (BORDER) bacc[i] += a[0] * wb0 + a[1] * wb1 + a[2] * wb2 + a[3] * wb3;
    #pragma unroll
                    for (int j = i; j < NFB; ++j) {
                        const int t = tile_w<NFB>(i, j), tw = tile_w<NFB>(p, j);
    #pragma unroll
                        for (int e = 0; e < 4; ++e) acc[t] = WMF_MFMA16(a[e], acc[tw][e], acc[t]);
                    }
                    if (!RELANE && i == p + 1) Xnext = invert(p + 1); // tile (p + 1, p + 1) is final: look ahead
                }
            }
        }
    }
    // The backward pass reads what OTHER lanes (r == 0) stored into Wv / Wb above.  Per thread nothing orders a lane's load
    // after another lane's store, and hipcc did move the last pivot's load above the store for the lanes that do not
    // write; the wave-scope fence pair and the scheduling barrier pin the order the wave's lockstep execution relies on.
    if constexpr (WASM) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (!WREG) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // ---- D: g_p = w_p - sum_{j > p} W_pj g_j ; gb[j] = g_j[lane & 15] on every lane
    tb = 0.f;                                                // BORDER: the last unknown
    if constexpr (BORDER) {
        cacc = wmf_qsum(cacc + wmf_row16_sum(cacc2));
        eacc = wmf_qsum(eacc + wmf_row16_sum(eacc2));
        const float piv = diag + cacc;                      // identity + c - sum_p b_p^T w^b_p
        if (!(piv > 1e-20f)) ok = false;
        tb = eacc * __builtin_amdgcn_rcpf(piv);
    }
    if (!WMF_ABL(dbg, 1)) {
#pragma unroll
        for (int p = NFB - 1; p >= 0; --p) {
            relane();
            float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = p + 1; j < NFB; ++j) {
                const int t = tile_w<NFB>(p, j);
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) s[reg] += (F16T ? -acc[t][reg] : acc[t][reg]) * gb[j];      // (F16T: the tile holds -W_pj)
            }
            if (p + 1 < NFB) wmf_row16_sum4(s[0], s[1], s[2], s[3]);
            float gsel = 0.f;
            f32x4 wl = f32x4{0.f, 0.f, 0.f, 0.f}, wbl = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (WASM) { wl = lds_get4(wv_lds, p); if constexpr (BORDER) wbl = lds_get4(wb_lds, p); }
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                float w;                                                              // w_p[4q + reg]
                if constexpr (WREG) { w = wvs[p][reg]; if constexpr (BORDER) w -= tb * wbs[p][reg]; }
                else if constexpr (WASM) { w = wl[reg]; if constexpr (BORDER) w -= tb * wbl[reg]; }
                else { w = Wv[16 * p + 4 * q + reg]; if constexpr (BORDER) w -= tb * Wb[16 * p + 4 * q + reg]; }
                const float gv = w - s[reg];                                         // g_p[4q + reg] on every lane (., q)
                gsel = ((r & 3) == reg) ? gv : gsel;
            }
            gb[p] = wmf_fetch_own_group(gsel);             // g_p[r] sits in q-group r >> 2, in the lanes whose r & 3 matches
        }
    }
    if constexpr (!GJ_LDS) {
        // the lean sweep's pivot test lets a NaN through (wmf_common.h): it is in the solution now
        bool bad = !(pmin > WMF_PIVOT_MIN_BITS);
        // a tile with pivots far apart, or above WMF_PIVOT_CAP (rows that mix confidence weights far above the usual range with
        // ordinary ones): the row goes to the pivoted LU kernel (wmf_common.h, gj_inv_step)
        bad |= pmax > WMF_PIVOT_CAP_BITS || spread > WMF_PIVOT_SPREAD_BITS;
#pragma unroll
        for (int p = 0; p < NFB; ++p) bad |= !(fabsf(gb[p]) < 3.0e38f);
        if constexpr (BORDER) bad |= !(fabsf(tb) < 3.0e38f);
        if (__any(bad)) ok = false;
    }
}
