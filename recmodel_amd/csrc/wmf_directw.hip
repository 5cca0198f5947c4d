// Heavy rows for 64 < f <= 144 (k = 128 with or without biases): ONE WAVE PER ROW, the whole f x f
// system lives in MFMA accumulator registers, no workgroup barriers (gfx950).
//
//   g_u = (I + V_u^T D V_u)^-1 V_u^T p          (RecModel/wmf_model.py:233-239 in whitened coordinates)
//
// Tile (bi, bj), bi <= bj, of B = I + V^T D V is one 16 x 16 accumulator: lane (r = l & 15, q = l >> 4)
// holds B[16 bi + 4q + reg][16 bj + r].  The right-hand side never becomes a tile: f32 MFMA and the VALU share one
// pipe on gfx950, and a vector riding in a 16 x 16 tile would use 1/16 of each MFMA.  It is
// kept as NFB registers y[fb] = (V^T p)[16 fb + lane & 15] and updated with plain FMAs.
//   A. entries stream from HBM straight into operand registers (lane loads V[idx_{4s+q}][16 fb + r]),
//      one 16-entry group ahead:  tile(bi,bj) += frag[bi]^T (w frag[bj]),  y[bi] += p frag[bi].
//   C. block elimination without square roots (B = U^T D U, D_p = the pivot blocks):
//        X   = inverse of the diagonal tile -- a symmetric tile in accumulator layout IS the row-distributed
//              layout of the Gauss-Jordan sweep of wmf_solve.hip, so it is inverted in place with 16
//              v_fmac_dpp steps, and the result is already the MFMA A operand;
//        W_pj = X B_pj  : four MFMAs with the tile's own registers as the B operand (k = 4q + reg);
//        B_ij -= B_pi^T W_pj : both operands are registers the lane already holds (element e of the original tile
//              (p,i) and of W_pj in accumulator layout are exactly what MFMA step e wants) -- no LDS panels;
//        w_p  = X y_p (DPP row sums), y_i -= B_pi^T w_p with the A operands the MFMAs just loaded.
//   D. g_p = w_p - sum_{j>p} W_pj g_j with DPP row sums; no triangular solves are left.
// A non-positive pivot (system not positive definite: possible with biases) bounces the row to the
// pivoted LU kernel.
//
// BORDER (f = 16 NFB + 1: k a multiple of 16 plus the bias column).  A ninth block column would hold ONE real
// column and cost 9 of 45 tiles in every k-step; instead the last feature is a border of the NFB-block system,
//     [ A   b ] [ g ]   [ y ]          A, y: the tiles and the vector above,  b = column f-1,  c = B[f-1][f-1],
//     [ b^T c ] [ t ] = [ e ]
// b rides through the elimination exactly like y (w^b_p = X_p b_p, b_i -= B_pi^T w^b_p), the last pivot is the
// scalar  c - sum_p b_p^T w^b_p,  t = (e - sum_p b_p^T w^y_p) / pivot,  and the backward pass starts from
// w^y_p - t w^b_p.  Measured at k = 128 + bias: the f = 128 kernel runs 38 ms where the nine-block one ran 50.
#include "wmf_common.h"
#include "wmf_internal.h"
#include "wmf_stream.h"
#include "wmf_dw_elim.h"

#include <type_traits>

#include <utility>

// Tunables per factor width (measured on MI355X, tools/kernel_lab.py): k-steps per pipelined group, ring
// depth, waves per SIMD the register budget is cut for.
#ifndef WMF_DW_GS
#define WMF_DW_GS 2
#endif
#ifndef WMF_DW_DEPTH
#define WMF_DW_DEPTH 3
#endif
#ifndef WMF_DW_DEPTH_WIDE
#define WMF_DW_DEPTH_WIDE 3
#endif
#ifndef WMF_DW_GS_WIDE
#define WMF_DW_GS_WIDE 2
#endif
#ifndef WMF_DW_OCC8
#define WMF_DW_OCC8 1
#endif
#ifndef WMF_DW_OCC4
#define WMF_DW_OCC4 3
#endif
#ifndef WMF_DW_GJ_LDS
#define WMF_DW_GJ_LDS -1          // multiplier column of the tile inverse: 1 = ds_bpermute, 0 = two VALU lane swaps, -1 = by occupancy
#endif
template <int NFB>
struct DwCfg {
    static constexpr int GS = (NFB <= 4) ? WMF_DW_GS : WMF_DW_GS_WIDE;
    static constexpr int DEPTH = (NFB <= 4) ? WMF_DW_DEPTH : WMF_DW_DEPTH_WIDE;
    static constexpr int OCC = NFB <= 4 ? WMF_DW_OCC4 : (NFB <= 6 ? 2 : (NFB <= 8 ? WMF_DW_OCC8 : 1));      // waves per SIMD
};


// MODE 0: one wave per row (accumulate + eliminate).
// Rows with more than WMF_HEAVY_T entries are split (power-law degrees):
// MODE 1: one wave per SEGMENT of such a row: accumulate its WMF_SEG entries, store the partial tiles;
// MODE 2: one wave per heavy row: add the partial tiles of its segments in order, then eliminate.
// floats per segment in the partial buffer of split rows: the tiles in tile_w order -- an off-diagonal tile as
// [reg][lane] (256 floats), a diagonal tile as its upper triangle only (136 floats, element (row <= col) at
// col (col + 1) / 2 + row: partial systems cross xGMI in reduce mode, and the lower triangle is the same numbers) --
// then y [fb][lane], then (BORDER) b [fb][lane], c [lane], e [lane]
// (WMF_DW_TRI, WMF_DW_TILES, WMF_DW_PARTIAL, tile_off: wmf_dw_elim.h -- the LDS-DMA kernel writes the same layout)

template <int NFB, int MODE, bool BORDER>
__global__ __launch_bounds__(64, DwCfg<NFB>::OCC) void solve_directw_kernel(const int32_t* __restrict__ rows, int64_t count,
                                                              const float* __restrict__ V, const float* __restrict__ side,
                                                              const int64_t* __restrict__ indptr,
                                                              const int32_t* __restrict__ indices,
                                                              const float* __restrict__ vals, int f, int ld,
                                                              float* __restrict__ g, int32_t* __restrict__ fb_rows,
                                                              int32_t* __restrict__ fb_count, int dbg,
                                                              const int64_t* __restrict__ seg_lo, const int32_t* __restrict__ seg_d,
                                                              const int32_t* __restrict__ seg_first, float* __restrict__ partial,
                                                              int slot_a, int slot_b, const int32_t* __restrict__ count_dev) {
    if (count_dev) count = *count_dev;                           // (the rows the iteration kernel bounced: the count is on the device)
    constexpr int NT = NFB * (NFB + 1) / 2;
    constexpr int GS = DwCfg<NFB>::GS;                           // MFMA k-steps (4 entries each) per pipelined group
    __shared__ __attribute__((aligned(16))) float Wv[NFB * 16];              // w_p = X_p y_p, kept for the backward pass
    __shared__ __attribute__((aligned(16))) float Wb[BORDER ? NFB * 16 : 4]; // w^b_p = X_p b_p
    const int lane = threadIdx.x;
    const int r = lane & 15, q = lane >> 4;
    int baddr[4];
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) baddr[kq] = (r + 16 * kq) * 4;
    // side != NULL (BORDER only): the split layout of a bias model's fixed side (wmf_internal.h) -- V holds packed body rows
    // of ldv = f - 1 floats, side the {last feature, bias} pairs; ld stays the stride of g
    const int ldv = (BORDER && side) ? f - 1 : ld;

    // Row pipeline (wmf_stream.h): factor rows are requested DEPTH groups ahead; the next row's first loads
    // are requested before this row's elimination starts.
    constexpr int DEPTH = DwCfg<NFB>::DEPTH;
    using Stream = WmfRowStream<NFB + (BORDER ? 1 : 0), GS, DEPTH>;   // BORDER: one more (dword) block, lane r = 0 of it is the border feature
    Stream st;
    int u = 0, d = 0;
    int64_t lo = 0;
    int64_t it = blockIdx.x;
    auto item = [&](int64_t i, int& u_, int64_t& lo_, int& d_) {   // work item i: a row (MODE 0, 2) or a segment (MODE 1)
        if constexpr (MODE == 1) { u_ = 0; lo_ = seg_lo[i]; d_ = seg_d[i]; }
        else if constexpr (MODE == 2) { u_ = rows ? rows[i] : (int)i; lo_ = 0; d_ = 0; }     // rows == NULL: row i itself
        else { u_ = rows[i]; lo_ = indptr[u_]; d_ = (int)(indptr[u_ + 1] - lo_); }
    };
    if (it < count) item(it, u, lo, d);
    auto prime = [&](int64_t lo_, int d_) {
        if constexpr (MODE == 2) return;
        st.load_block(0, lo_, d_, indices, vals, lane, 0);
        st.load_block(1, lo_, d_, indices, vals, lane, 1);
        st.fetch_meta(0, q);
        [&]<int... Ss>(std::integer_sequence<int, Ss...>) {
            (st.template load_group<Ss>(Ss, V, ldv, r, q, BORDER ? side : nullptr), ...);
        }(std::make_integer_sequence<int, DEPTH>{});
    };
    if (it < count) prime(lo, d);

    for (; it < count; it += gridDim.x) {
        const int ngroups = (MODE == 2) ? 0 : (d + Stream::EPG - 1) / Stream::EPG;
        const int64_t itn = it + gridDim.x;
        int un = 0, dn = 0;
        int64_t lon = 0;
        if (itn < count) item(itn, un, lon, dn);

        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        float racc[NFB], bacc[BORDER ? NFB : 1];
        float cacc = 0.f, eacc = 0.f;                            // border: c and e (this lane's q share; equal across r)
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) racc[fb] = 0.f;
#pragma unroll
        for (int fb = 0; fb < (BORDER ? NFB : 1); ++fb) bacc[fb] = 0.f;

        // ---- A (the rhs V^T p rides on the VALU and becomes block column NFB afterwards)
        auto step = [&](auto slot, int G) {
            constexpr int S = decltype(slot)::value;
            if (G >= ngroups) return;
            if (!WMF_ABL(dbg, 2)) {
#pragma unroll
                for (int t = 0; t < GS; ++t) {
                    float fw[NFB];
                    st.template mask_tail<S>(t, ld, r);          // (the row-major width: in the split layout the pair block has no tail)
                    float wt = st.w[S][t], pt = st.p[S][t];
                    if constexpr (BORDER) {
                        // split layout: the dword block that brings the border feature to lane r = 0 brings the fixed side's
                        // bias to lane r = 1
                        if (side) {
                            const bool real = Stream::EPG * G + 4 * t + q < d;
                            const float bs = real ? wmf_dpp<0x151>(st.fr[S][t][NFB]) : 0.f;     // row_newbcast:1
                            wt -= bs;
                            pt -= bs;
                        }
                    }
#pragma unroll
                    for (int fb = 0; fb < NFB; ++fb) { fw[fb] = st.fr[S][t][fb] * wt; racc[fb] += st.fr[S][t][fb] * pt; }
                    if constexpr (BORDER) {
                        const float bf = wmf_dpp<0x150>(st.fr[S][t][NFB]);          // border feature of this lane's entry (row_newbcast:0)
                        const float bw = bf * wt;
#pragma unroll
                        for (int fb = 0; fb < NFB; ++fb) bacc[fb] += st.fr[S][t][fb] * bw;
                        cacc += bf * bw;
                        eacc += bf * pt;
                    }
                    int tt = 0;
#pragma unroll
                    for (int bi = 0; bi < NFB; ++bi) {
#pragma unroll
                        for (int bj = bi; bj < NFB; ++bj, ++tt) acc[tt] = WMF_MFMA16(st.fr[S][t][bi], fw[bj], acc[tt]);
                    }
                }
            }
            const int next = G + DEPTH;
            if (next < ngroups) {
                if (next % Stream::GPB == 0) {
                    const int c = next / Stream::GPB;
                    st.load_block(c + 1, lo, d, indices, vals, lane, (c + 1) & 1);
                }
                st.template load_group<S>(next, V, ldv, r, q, BORDER ? side : nullptr);
            }
        };
        for (int G0 = 0; G0 < ngroups; G0 += DEPTH) {
            [&]<int... Ss>(std::integer_sequence<int, Ss...>) {
                (step(std::integral_constant<int, Ss>{}, G0 + Ss), ...);
            }(std::make_integer_sequence<int, DEPTH>{});
        }
        // racc[fb] = this lane's share (its q) of y[16 fb + r]; the four shares are added when block fb becomes the pivot
        if (itn < count) prime(lon, dn);                         // next row's first loads fly during the elimination
        if constexpr (MODE == 1) {                               // partial tiles of this segment: [tile][reg][lane]
            float* out = partial + (it * slot_a + slot_b) * (int64_t)WMF_DW_PARTIAL(NFB, BORDER);   // slot of work item `it`
#pragma unroll
            for (int bi = 0; bi < NFB; ++bi) {
#pragma unroll
                for (int bj = bi; bj < NFB; ++bj) {
                    const int t = tile_w<NFB>(bi, bj);
                    float* to = out + tile_off<NFB>(bi, bj);
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        if (bi != bj) to[reg * 64 + lane] = acc[t][reg];
                        else if (4 * q + reg <= r) to[r * (r + 1) / 2 + 4 * q + reg] = acc[t][reg];   // element (4q + reg, r)
                    }
                }
            }
            float* vo = out + WMF_DW_TILES(NFB);
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) vo[fb * 64 + lane] = racc[fb];
            if constexpr (BORDER) {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) vo[(NFB + fb) * 64 + lane] = bacc[fb];
                vo[2 * NFB * 64 + lane] = cacc;
                vo[(2 * NFB + 1) * 64 + lane] = eacc;
            }
            u = un; lo = lon; d = dn;
            continue;
        }
        if constexpr (MODE == 2) {                               // sum the segments of heavy row `it` in a fixed order
            // seg_first == NULL: row `it` owns the slot_a consecutive slots it * slot_a ..; seg_first with slot_a == 0: the
            // segments were already summed into the row's first slot (wmf_launch_combine_segments)
            const int64_t sg0 = seg_first ? seg_first[it] : it * slot_a;
            const int64_t sg1 = seg_first ? (slot_a == 0 ? sg0 + 1 : seg_first[it + 1]) : sg0 + slot_a;
            for (int64_t sgm = sg0; sgm < sg1; ++sgm) {
                const float* in = partial + sgm * (int64_t)WMF_DW_PARTIAL(NFB, BORDER);
#pragma unroll
                for (int bi = 0; bi < NFB; ++bi) {
#pragma unroll
                    for (int bj = bi; bj < NFB; ++bj) {
                        const int t = tile_w<NFB>(bi, bj);
                        const float* ti = in + tile_off<NFB>(bi, bj);
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            const int row = 4 * q + reg, lo_ = min(row, r), hi_ = max(row, r);
                            acc[t][reg] += (bi != bj) ? ti[reg * 64 + lane] : ti[hi_ * (hi_ + 1) / 2 + lo_];
                        }
                    }
                }
                const float* vi = in + WMF_DW_TILES(NFB);
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) racc[fb] += vi[fb * 64 + lane];
                if constexpr (BORDER) {
#pragma unroll
                    for (int fb = 0; fb < NFB; ++fb) bacc[fb] += vi[(NFB + fb) * 64 + lane];
                    cacc += vi[2 * NFB * 64 + lane];
                    eacc += vi[(2 * NFB + 1) * 64 + lane];
                }
            }
        }

        // ---- C, D: block elimination and backward pass (wmf_dw_elim.h)
        bool ok = true;
        float gb[NFB];
        float tb = 0.f;
        {
            constexpr bool GJ_LDS = WMF_DW_GJ_LDS < 0 ? (DwCfg<NFB>::OCC > 1) : (WMF_DW_GJ_LDS != 0);
            dw_eliminate<NFB, BORDER, GJ_LDS>(acc, racc, bacc, cacc, eacc, Wv, Wb, r, q, baddr, dbg, gb, tb, ok);
        }
        if (!ok) {
            if (lane == 0) fb_rows[atomicAdd(fb_count, 1)] = u;
        } else if (q == 0) {
#pragma unroll
            for (int p = 0; p < NFB; ++p) {
                const int c = Stream::real_col(p, r);            // undo the feature permutation of the row stream
                if (c < ld) g[(int64_t)u * ld + c] = (c < f) ? gb[p] : 0.f;
            }
            if constexpr (BORDER) {
                const int c = 16 * NFB + r;                      // the border column and the padding behind it
                if (c < ld) g[(int64_t)u * ld + c] = (r == 0) ? tb : 0.f;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        u = un; lo = lon; d = dn;
    }
}

template <int NFB, bool BORDER>
static void launch_directw_nfb(const wmf_plan* pl, const float* V, const float* side, const int64_t* indptr,
                               const int32_t* indices, const float* vals, int f, int ld, float* g, int dbg, hipStream_t st) {
    constexpr int waves_per_cu = 4 * DwCfg<NFB>::OCC;
    const int64_t cap = 256 * waves_per_cu * 3;                  // resident waves, three rounds queued
    const int32_t* rows = pl->rows[WMF_BIN_MFMA];
    const int64_t normal = pl->count[WMF_BIN_MFMA] - pl->heavy_count;
    // ROUND 4: the first iter_count of the normal rows (at most wmf_iter_dmax entries each, wmf_plan_create) go to the
    // matrix-free iteration kernel (wmf_iter.hip); what it cannot solve to float32 accuracy in a few applications of the
    // row's operator comes back as a device-side list and is eliminated below like every other row.  Debug flag 268435456
    // switches the iteration off (everything eliminated, as in round 3).
    const int64_t n_iter = wmf_iter_rows(pl, f, ld, side != nullptr);
    if (n_iter > 0)
        (void)wmf_launch_iter(rows, n_iter, V, side, indptr, indices, vals, f, ld, g, pl->iter_bounce_rows, pl->fallback_count + 1,
                              pl->iter_stats, pl->iter_info, st, (side && pl->rolled) ? 1 : 0);
    // two launches of the elimination kernel: the rows that were never candidates (count on the host), then the bounced ones
    // (count on the device; the grid is sized for the list's capacity and exits at once when the list is empty)
    for (int pass = 0; pass < 2; ++pass) {
        const int32_t* prow = pass ? pl->iter_bounce_rows : rows + n_iter;
        const int64_t pcount = pass ? n_iter : normal - n_iter;
        const int32_t* pdev = pass ? pl->fallback_count + 1 : nullptr;
        if (pcount <= 0) continue;
        // k = 128 with or without biases: the LDS-DMA ring kernel (wmf_directl.hip); debug flag 4096 keeps the register ring
        // and k = 64 since round 2: with the split-f16 accumulation AND elimination the LDS-DMA kernel, two waves per SIMD there,
        // takes 0.97 ms for cfg2's item side where the f32 register-ring kernel takes 1.39 (round 1, bf16 x 3 accumulation
        // and f32 elimination: 1.34 against 1.30; debug flag 65536 keeps the register ring at k = 64)
        // (side: NULL, or the {last feature, bias} pairs of the split layout, V then being the packed body)
        if (wmf_directl_supported(f, ld) && !(dbg & 4096) && (f >= 128 || !(dbg & 65536))) {
            (void)wmf_launch_directl(prow, pcount, V, side, indptr, indices, vals, f, ld, g, pl->fallback_rows, pl->fallback_count, st, pdev);
        } else {
            static const char* nm = wmf_kname("solve_directw_kernel<%d, 0, %s>", NFB, BORDER ? "true" : "false");
            static const char* nmb = wmf_kname("solve_directw_kernel<%d, 0, %s> [bounced]", NFB, BORDER ? "true" : "false");
            WMF_LAUNCH(pass ? nmb : nm, (solve_directw_kernel<NFB, 0, BORDER>), dim3((unsigned)(pcount < cap ? pcount : cap)), dim3(64), 0, st,
                       prow, pcount, V, side, indptr, indices, vals, f, ld, g, pl->fallback_rows, pl->fallback_count, dbg,
                       nullptr, nullptr, nullptr, nullptr, 1, 0, pdev);
        }
    }
    if (pl->heavy_count > 0) {
        const int64_t nseg = pl->seg_total;
        static const char* nm1 = wmf_kname("solve_directw_kernel<%d, 1, %s>", NFB, BORDER ? "true" : "false");
        static const char* nm2 = wmf_kname("solve_directw_kernel<%d, 2, %s>", NFB, BORDER ? "true" : "false");
        // (k = 128: the segments through the LDS-DMA kernel as well -- same partial layout; debug flags 4096 / 16777216: here)
        if (NFB == 8 && wmf_directl_supported(f, ld) && !(dbg & (4096 | 8192 | 16777216))) {
            (void)wmf_launch_directl_segments(nseg, V, side, indices, vals, f, ld, pl->seg_lo, pl->seg_d, pl->partial, st);
        } else
        WMF_LAUNCH(nm1, (solve_directw_kernel<NFB, 1, BORDER>), dim3((unsigned)(nseg < cap ? nseg : cap)), dim3(64), 0, st, rows,
                   nseg, V, side, indptr, indices, vals, f, ld, g, pl->fallback_rows, pl->fallback_count, dbg, pl->seg_lo,
                   pl->seg_d, pl->seg_first, pl->partial, 1, 0, (const int32_t*)nullptr);
        const int64_t nh = pl->heavy_count;
        wmf_launch_combine_segments(pl, WMF_DW_PARTIAL(NFB, BORDER), st);
        WMF_LAUNCH(nm2, (solve_directw_kernel<NFB, 2, BORDER>), dim3((unsigned)(nh < cap ? nh : cap)), dim3(64), 0, st,
                   rows + normal, nh, V, side, indptr, indices, vals, f, ld, g, pl->fallback_rows, pl->fallback_count, dbg,
                   pl->seg_lo, pl->seg_d, pl->seg_first, pl->partial, 0, 0, (const int32_t*)nullptr);
    }
}

// Partial systems for the reduce-scatter exchange (engine.py, "reduce mode"): MODE 1 over every row of a CSR (one
// segment per row, slot = row) and MODE 2 over a buffer of summed partial systems (one slot per row, row = slot).
template <int NFB, bool BORDER>
static void launch_accumulate_nfb(const float* V, const float* side, const int64_t* indptr, const int32_t* degrees, const int32_t* indices,
                                  const float* vals, int64_t n, int f, int ld, float* partial, int slot_stride, int slot_offset,
                                  hipStream_t st) {
    const int64_t cap = 256 * 4 * DwCfg<NFB>::OCC * 3;
    static const char* nm = wmf_kname("solve_directw_kernel<%d, 1, %s>", NFB, BORDER ? "true" : "false");
    WMF_LAUNCH(nm, (solve_directw_kernel<NFB, 1, BORDER>), dim3((unsigned)(n < cap ? n : cap)), dim3(64), 0, st, nullptr, n, V,
               side, indptr, indices, vals, f, ld, nullptr, nullptr, nullptr, wmf_debug_flags & ~3, indptr, degrees,
               nullptr, partial, slot_stride, slot_offset, (const int32_t*)nullptr);
}
template <int NFB, bool BORDER>
static void launch_eliminate_nfb(float* partial, int64_t n, int slots_per_row, int f, int ld, float* g, int32_t* fb_rows,
                                 int32_t* fail_count, hipStream_t st) {
    const int64_t cap = 256 * 4 * DwCfg<NFB>::OCC * 3;
    static const char* nm = wmf_kname("solve_directw_kernel<%d, 2, %s>", NFB, BORDER ? "true" : "false");
    WMF_LAUNCH(nm, (solve_directw_kernel<NFB, 2, BORDER>), dim3((unsigned)(n < cap ? n : cap)), dim3(64), 0, st, nullptr, n,
               nullptr, nullptr, nullptr, nullptr, nullptr, f, ld, g, fb_rows, fail_count, wmf_debug_flags & ~3, nullptr,
               nullptr, nullptr, partial, slots_per_row, 0, (const int32_t*)nullptr);
}
static bool dw_border(int f) { return wmf_dw_border(f); }

int64_t wmf_directw_partial_floats(int f) {
    if (f < 1 || f > 144) return 0;
    const int64_t nfb = dw_border(f) ? f / 16 : (f + 15) / 16;
    return nfb * (nfb - 1) / 2 * 256 + nfb * WMF_DW_TRI + (nfb + (dw_border(f) ? nfb + 2 : 0)) * 64;
}

int wmf_launch_accumulate(const float* V, const float* side, const int64_t* indptr, const int32_t* degrees, const int32_t* indices,
                          const float* vals, int64_t n, int f, int ld, float* partial, int slot_stride, int slot_offset,
                          hipStream_t st) {
    if (n <= 0) return 0;
    if (f > 144) return -1;
    if (dw_border(f)) {
        switch (f / 16) {
#define C_(N) case N: launch_accumulate_nfb<N, true>(V, side, indptr, degrees, indices, vals, n, f, ld, partial, slot_stride, slot_offset, st); break;
            C_(1) C_(2) C_(4) C_(5) C_(6) C_(8)
#undef C_
            default: return -1;
        }
        return 0;
    }
    switch ((f + 15) / 16) {
#define C_(N) case N: launch_accumulate_nfb<N, false>(V, nullptr, indptr, degrees, indices, vals, n, f, ld, partial, slot_stride, slot_offset, st); break;
        C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9)
#undef C_
        default: return -1;
    }
    return 0;
}

int wmf_launch_eliminate(float* partial, int64_t n, int slots_per_row, int f, int ld, float* g, int32_t* fb_rows,
                         int32_t* fail_count, hipStream_t st) {
    if (n <= 0) return 0;
    if (f > 144) return -1;
    if (dw_border(f)) {
        switch (f / 16) {
#define C_(N) case N: launch_eliminate_nfb<N, true>(partial, n, slots_per_row, f, ld, g, fb_rows, fail_count, st); break;
            C_(1) C_(2) C_(4) C_(5) C_(6) C_(8)
#undef C_
            default: return -1;
        }
        return 0;
    }
    switch ((f + 15) / 16) {
#define C_(N) case N: launch_eliminate_nfb<N, false>(partial, n, slots_per_row, f, ld, g, fb_rows, fail_count, st); break;
        C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9)
#undef C_
        default: return -1;
    }
    return 0;
}

int wmf_launch_directw(const wmf_plan* pl, const float* V, const float* side, const int64_t* indptr,
                       const int32_t* indices, const float* vals, int f, int ld, float* g, hipStream_t st) {
    if (pl->count[WMF_BIN_MFMA] <= 0) return 0;
    const int dbg = wmf_debug_flags;
    // k = 16 m with biases: m blocks and a border column.  The row stream must deliver the border feature as its own
    // dword block (lane 0), which it does unless m + 1 is a multiple of 4 (then all blocks are 16-byte pieces).
    if (dw_border(f)) {
        switch (f / 16) {
#define C_(N) case N: launch_directw_nfb<N, true>(pl, V, side, indptr, indices, vals, f, ld, g, dbg, st); break;
            C_(1) C_(2) C_(4) C_(5) C_(6) C_(8)
#undef C_
            default: return -1;
        }
        return 0;
    }
    switch ((f + 15) / 16) {
#define C_(N) case N: launch_directw_nfb<N, false>(pl, V, side, indptr, indices, vals, f, ld, g, dbg, st); break;
        C_(1) C_(2) C_(3) C_(4) C_(5) C_(6) C_(7) C_(8) C_(9)
#undef C_
        default: return -1;
    }
    return 0;
}
