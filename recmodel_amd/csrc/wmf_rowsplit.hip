// Rows with more than 32 stored entries when 144 < f <= 257 (k = 256 with or without biases, and the widths between):
// the f x f whitened system  (I + V_u^T D V_u) g = V_u^T p  is too big for one wave's accumulators
// (wmf_directw.hip), so FOUR waves share it BY BLOCK ROWS.  Reference arithmetic: RecModel/wmf_model.py:233-239.
//
// Tile (bi, bj), bi <= bj < NFB <= 16, of the upper triangle lives in the accumulators of the wave that owns block
// row bi.  Wave w owns rows {w, 7 - w, 8 + w, 15 - w}: 34 tiles each for NFB = 16 (136 accumulator registers), and
// because w is a compile-time constant of the code a wave runs (four specialised copies of the row code, selected
// once per workgroup), every accumulator index AND every LDS offset is an immediate:
//   A. entries are staged 16 at a time through two LDS buffers (one barrier per chunk); per k-step a wave reads the NFB fragments once (they are the
//      B operands of all its tiles and, for its own four rows, the A operands) and issues its 34 MFMAs back to
//      back:  tile(bi, bj) += frag[bi]^T (w frag[bj]);  the right-hand side y[bi] += p frag[bi] stays on the VALU.
//   C. block LDL^T as in wmf_directw.hip: the owner of row p inverts tile (p, p) in its registers (16 DPP
//      Gauss-Jordan steps), forms W_pj = X B_pj for its whole row without any exchange, publishes the original row
//      and the W row to two LDS panels, and after a barrier every wave updates its own rows, B_ij -= B_pi^T W_pj
//      and y_i -= B_pi^T w_p (w_p = X y_p); a second barrier frees the panels.  Two workgroups share a CU, so one
//      computes while the other waits for its pivot.
//   D. g_p = w_p - sum_{j>p} W_pj g_j: the owner of row p has every W_pj in registers; rows are finished from the
//      last to the first, one barrier each.
// BORDER (f = 16 NFB + 1, k = 16 NFB with biases): the last column is a border of the NFB-block system exactly as in
// wmf_directw.hip -- a second right-hand side through the elimination, a scalar last pivot.
// A non-positive pivot bounces the row to the pivoted LU kernel (wmf_wide.hip).  Compared with the run-time-indexed
// kernel in wmf_wide.hip (still used for 257 < f <= 272): no LDS operand reads inside the MFMA stream beyond the shared
// fragments, no register or scalar spills, two barriers per pivot instead of three.
#include "wmf_common.h"
#include "wmf_internal.h"

#include <utility>

#ifndef RS_GJ_LDS
#define RS_GJ_LDS true   /* multiplier column of the tile inverse by ds_bpermute (true) or by VALU lane swaps (false) */
#endif

typedef _Float16 rs_f16x8 __attribute__((ext_vector_type(8)));
// a tile's four values per lane (k = 4q + e) as { h_0..h_3, l_0..l_3 }: the "natural" split-f16 operand of wmf_dw_elim.h
__device__ __forceinline__ rs_f16x8 rs_split_natural(const f32x4 v) {     // (C++ on purpose: see dw_split_natural, wmf_dw_elim.h)
    rs_f16x8 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma clang fp contract(off)
        const _Float16 h = (_Float16)v[e];
        o[e] = h;
        o[4 + e] = (_Float16)(v[e] - (float)h);
    }
    return o;
}
__device__ __forceinline__ rs_f16x8 rs_dup_hi(const rs_f16x8 n) { return __builtin_shufflevector(n, n, 0, 1, 2, 3, 0, 1, 2, 3); }
__device__ __forceinline__ rs_f16x8 rs_dup_lo(const rs_f16x8 n) { return __builtin_shufflevector(n, n, 4, 5, 6, 7, 4, 5, 6, 7); }

template <int NFB, bool BORDER = false, bool F16A = false>
struct RsCfg {
    static constexpr int NW = 4, NTHR = 256;
    static constexpr int FP = 16 * NFB;
    // row stride = 16 (mod 32): the two k rows of a half wave hit disjoint banks; BORDER needs room for column FP
    static constexpr int LDV = (FP % 32 == 16) ? (BORDER ? FP + 32 : FP) : FP + 16;
    static constexpr int RC = 16;                                 // staged entries per chunk
    static constexpr int PF = (RC * (FP / 4 + (BORDER ? 1 : 0)) + NTHR - 1) / NTHR;  // 16-byte pieces prefetched per thread and chunk
    // LDS carve (floats)
    static constexpr int OFF_VS = 0;                              // [2][RC][LDV] staged factor rows, double buffered
    // F16A: instead the split operands of ONE 32-entry chunk: high parts [NFB][4][16] x 8 halves, then the low parts
    // Each part is SIXTEEN blocks whatever NFB is (round 4): every lane of a wave stages its piece -- lanes past the last
    // tile column (4 NFB .. 63 at NFB < 16) write blocks nobody reads -- so that the staging has no divergent region
    // (see stage32).  +3 KB per part at NFB = 13.
    static constexpr int OPS = 16 * 64 * 4;                       // floats per part (16 bytes per (block, q, r))
    static constexpr int OFF_W = OFF_VS + (F16A ? 2 * OPS : 2 * RC * LDV);           // [2][RC] weights
    static constexpr int OFF_P = OFF_W + 2 * RC;                  // [2][RC] w + 1 (0 past the end of the row)
    static constexpr int OFF_PAN = OFF_P + 2 * RC;                 // [2][NFB][16][20]: {originals, W} of the pivot row, by block column
    // (F16A: the same two panels as split-f16 operands, [2][NFB][64 lanes] x 16 bytes; at least the 10 FP + 8 floats of the
    // right-hand-side scratch)
    // (right-hand-side scratch: two planes [4 waves][256] for y and b -- a plane row is 64 lanes x 4 floats, written by every
    // lane --, then y [FP], b [FP], the waves' c / e [8])
    static constexpr int RHS = 2 * 4 * 256 + 2 * FP + 8;
    static constexpr int PAN = F16A ? (2 * NFB * 256 > RHS ? 2 * NFB * 256 : RHS) : 2 * NFB * 320;
    static constexpr int OFF_WV = OFF_PAN + PAN;                  // [16] w_p of the pivot row (+ spare)
    static constexpr int OFF_G = OFF_WV + 32;                     // [FP] solution
    static constexpr int OFF_WB = OFF_G + FP;                     // BORDER: [FP] w^b_p of every pivot row, then [8] scalars
    static constexpr int OFF_FLAG = OFF_WB + (BORDER ? FP + 8 : 0);   // [4]
    static constexpr int TOTAL = OFF_FLAG + 4;
};

// block rows of wave W, in increasing order; a row >= NFB does not exist
template <int W> __device__ __host__ constexpr int rs_row(int s) { return s == 0 ? W : (s == 1 ? 7 - W : (s == 2 ? 8 + W : 15 - W)); }
// first accumulator of row slot s: rows hold NFB - row tiles each (columns row .. NFB - 1)
template <int NFB, int W> __device__ __host__ constexpr int rs_base(int s) {
    int b = 0;
    for (int t = 0; t < s; ++t) b += rs_row<W>(t) < NFB ? NFB - rs_row<W>(t) : 0;
    return b;
}
template <int NFB, int W> __device__ __host__ constexpr int rs_ntiles() { return rs_base<NFB, W>(4); }

// floats of one segment's partial system (MODE 1 -> MODE 2): the upper-triangle tiles as [tile_w][reg][lane], then y [FP],
// then (BORDER) b [FP] and the four waves' (c, e) shares
#define RS_PARTIAL(NFB, BORDER) (((NFB) * ((NFB) + 1) / 2) * 256 + 16 * (NFB) + ((BORDER) ? 16 * (NFB) + 8 : 0))
template <int NFB> __device__ __host__ constexpr int rs_tile_w(int bi, int bj) { return bi * NFB - (bi * (bi - 1)) / 2 + (bj - bi); }

// MODE 0: one workgroup per row (accumulate + eliminate).  Rows with more than WMF_HEAVY_T entries are split (SURVEY.md 7-E,
// power-law degrees; F16A only): MODE 1: one workgroup per SEGMENT: accumulate its entries, store the partial system;
// (wmf_launch_combine_segments adds the partial systems of a row's segments in order); MODE 2: one workgroup per heavy row:
// eliminate the summed system.
template <int NFB, int W, bool BORDER, bool F16A, int MODE>
__device__ __forceinline__ void rs_body(float* __restrict__ sm, const int32_t* __restrict__ rows, int64_t count,
                                        const float* __restrict__ V, const float* __restrict__ biasv,
                                        const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                        const float* __restrict__ vals, int f, int ld, float* __restrict__ g,
                                        int32_t* __restrict__ fb_rows, int32_t* __restrict__ fb_count, int dbg,
                                        const int64_t* __restrict__ seg_lo, const int32_t* __restrict__ seg_d,
                                        const int32_t* __restrict__ seg_first, float* __restrict__ partial) {
    static_assert(F16A || MODE == 0, "split rows: split-f16 kernel only");
    using C = RsCfg<NFB, BORDER, F16A>;
    constexpr int NT = rs_ntiles<NFB, W>();
    float* Wball = sm + C::OFF_WB;                               // BORDER only
    float* bsc = sm + C::OFF_WB + C::FP;                         // BORDER: [0] sum_p b_p^T w^b_p, [1] sum_p b_p^T w^y_p
    float* Vs = sm + C::OFF_VS; float* wsm = sm + C::OFF_W; float* psm = sm + C::OFF_P;
    float* Pan = sm + C::OFF_PAN; float* Wv = sm + C::OFF_WV; float* gs = sm + C::OFF_G;
    int* flag = reinterpret_cast<int*>(sm + C::OFF_FLAG);
    const int tid = threadIdx.x, lane = tid & 63;
    const int r = lane & 15, q = lane >> 4;
    const int nch = ld >> 2;
    int baddr[4];
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) baddr[kq] = (r + 16 * kq) * 4;

    // per-thread (row-in-chunk, piece) of each prefetched 16-byte piece; fixed across chunks
    int pj[C::PF], pc[C::PF];
#pragma unroll
    for (int i = 0; i < C::PF; ++i) { const int e = tid + C::NTHR * i; pj[i] = e / nch; pc[i] = e % nch; }
    float4 pre[C::PF];                                           // ONE staged chunk in flight (registers are what limits two
    float wpre = 0.f;                                            // workgroups per CU): requested while the previous one is consumed
    auto load_chunk = [&](int64_t lo_, int d_, int base) {
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
    if constexpr (!F16A) {
        for (int e = tid; e < 2 * C::RC * C::LDV; e += C::NTHR) Vs[e] = 0.f;   // pad columns [ld, LDV) stay zero for good
    }
    // ---- F16A: split-f16 accumulation (the scheme of wmf_directl.hip: operands scaled by sqrt(w) and split into f16 high
    // and low parts, three v_mfma_f32_16x16x32_f16 per tile and 32 entries instead of eight f32 MFMAs at twice the cycles).
    // The split is done ONCE per value, by the thread that stages it: wave W takes the entries 8W .. 8W + 7 of a 32-entry
    // chunk (the K slots of MFMA lane group q = W), lane c the 16-byte piece c of their rows, and writes the eight halves of
    // a feature as one 16-byte LDS word where the consumer lane (r, q) of EVERY wave reads its operand with one
    // ds_read_b128: hi / lo [block][q][r][8].  Right-hand side, border column and the two border scalars are summed by the
    // staging threads from the raw f32 values and combined over the four waves once per row.
    rs_f16x8* ops_hi = reinterpret_cast<rs_f16x8*>(sm + C::OFF_VS);
    rs_f16x8* ops_lo = reinterpret_cast<rs_f16x8*>(sm + C::OFF_VS + C::OPS);
    constexpr int NPC = 4 * NFB;                                 // 16-byte pieces that hold tile features
    float4 pre8[F16A ? 8 : 1];
    float wpre8[F16A ? 8 : 1], ppre8[F16A ? 8 : 1], bfpre8[(F16A && BORDER) ? 8 : 1];
    float yreg[4] = {0.f, 0.f, 0.f, 0.f}, breg[4] = {0.f, 0.f, 0.f, 0.f}, creg = 0.f, ereg = 0.f;
    auto load_chunk32 = [&](int64_t lo_, int d_, int base) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = base + 8 * W + j;
            const bool on = e < d_;
            const int64_t at = on ? lo_ + e : lo_;               // (wave-uniform: the row has entries, lo_ is one of them)
            const int idx = indices[at];
            float wv = vals[at];
            if (biasv) wv -= biasv[idx];
            wpre8[j] = on ? wv : 0.f;
            ppre8[j] = on ? wv + 1.f : 0.f;
            const bool mine = lane < NPC && lane < nch;         // (f < 16 NFB: the last pieces of the tile columns lie beyond the row)
            const float m = (on && mine) ? 1.f : 0.f;
            const float4 v = reinterpret_cast<const float4*>(V + (int64_t)idx * ld)[mine ? lane : 0];
            pre8[j] = make_float4(v.x * m, v.y * m, v.z * m, v.w * m);
            if constexpr (BORDER) bfpre8[j] = on ? V[(int64_t)idx * ld + 16 * NFB] : 0.f;
        }
    };
    auto stage32 = [&]() {
        float sw[8], bw[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sw[j] = __builtin_amdgcn_sqrtf(wpre8[j]);            // (negative weight: NaN, caught by the pivot test)
            bw[j] = 0.f;
            if constexpr (BORDER) {
                bw[j] = bfpre8[j] * wpre8[j];
                creg += bfpre8[j] * bw[j];
                ereg += bfpre8[j] * ppre8[j];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            rs_f16x8 h, l;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = e == 0 ? pre8[j].x : (e == 1 ? pre8[j].y : (e == 2 ? pre8[j].z : pre8[j].w));
                yreg[e] += x * ppre8[j];
#ifndef RS_Y_PAIRED
                // Keeps hipcc's SLP vectoriser from pairing y with b in one v_pk_fma_f32 chain: with that pairing the BORDER kernels
                // at NFB = 13 .. 15 lost y's reproducibility at two workgroups per CU (round-3 note in launch_rowsplit_f).
                // RS_Y_PAIRED (lab, with RS_NO_ALONE and without -fno-slp-vectorize) brings the failure back.
                asm volatile("" : "+v"(yreg[e]));
#endif
                if constexpr (BORDER) breg[e] += x * bw[j];
                {
#pragma clang fp contract(off)
                    const float sx = x * sw[j];
                    const _Float16 hh = (_Float16)sx;
                    h[j] = hh;
                    l[j] = (_Float16)(sx - (float)hh);
                }
            }
            // (every lane stores: blocks NFB .. 15 of the padded planes are a dump -- no divergent region in the staging, so that
            // nothing is spilled under one EXEC mask and reloaded under another)
            const int ft = 4 * lane + e;
            ops_hi[((ft >> 4) * 4 + W) * 16 + (ft & 15)] = h;
            ops_lo[((ft >> 4) * 4 + W) * 16 + (ft & 15)] = l;
        }
    };

    int64_t it = blockIdx.x;
    int u = 0, d = 0;
    int64_t lo = 0;
    auto item = [&](int64_t i, int& u_, int64_t& lo_, int& d_) {     // work item i: a row (MODE 0, 2) or a segment (MODE 1)
        if constexpr (MODE == 1) { u_ = 0; lo_ = seg_lo[i]; d_ = seg_d[i]; }
        else if constexpr (MODE == 2) { u_ = rows[i]; lo_ = 0; d_ = 0; }
        else { u_ = rows[i]; lo_ = indptr[u_]; d_ = (int)(indptr[u_ + 1] - lo_); }
    };
    if (it < count) {
        item(it, u, lo, d);
        if constexpr (MODE != 2) { if constexpr (F16A) load_chunk32(lo, d, 0); else load_chunk(lo, d, 0); }
    }
    for (; it < count; it += gridDim.x) {
        const int nchunks = (d + C::RC - 1) / C::RC;
        const int64_t itn = it + gridDim.x;
        int un = 0, dn = 0;
        int64_t lon = 0;
        if (itn < count) item(itn, un, lon, dn);

        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        float yacc[4] = {0.f, 0.f, 0.f, 0.f};                    // this lane's q share of y[16 row + r], per row slot
        float bacc[4] = {0.f, 0.f, 0.f, 0.f};                    // BORDER: the same for the border column b
        float cacc = 0.f, eacc = 0.f;                            // BORDER: c and e, this lane's q share (every wave has them)
        if (tid == 0) { flag[0] = 0; if constexpr (BORDER) { bsc[0] = 0.f; bsc[1] = 0.f; } }

        // ---- A.  `pre` holds chunk 0 on entry (requested at the start, or during the previous row's elimination).
        //      Two LDS buffers: chunk c + 1 is written while slower waves may still read chunk c, so one barrier per
        //      chunk is enough, and the global loads of chunk c + 2 fly during the MFMAs of chunk c + 1.
        if constexpr (F16A) {
            const int nchunks32 = (d + 31) >> 5;
            for (int c = 0; c < nchunks32; ++c) {
                stage32();                                       // the previous chunk's operand reads ended at the barrier below
                __syncthreads();
                if (c + 1 < nchunks32) load_chunk32(lo, d, 32 * (c + 1));         // flies during this chunk's MFMAs
                else if (itn < count && MODE != 2) load_chunk32(lon, dn, 0);     // ... or during the elimination
                rs_f16x8 Ah[4], Al[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int bi = rs_row<W>(s);
                    if (bi < NFB) { Ah[s] = ops_hi[(bi * 4 + q) * 16 + r]; Al[s] = ops_lo[(bi * 4 + q) * 16 + r]; }
                }
#pragma unroll
                for (int bj = 0; bj < NFB; ++bj) {
                    const rs_f16x8 Bh = ops_hi[(bj * 4 + q) * 16 + r], Bl = ops_lo[(bj * 4 + q) * 16 + r];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int bi = rs_row<W>(s);
                        if (bi < NFB && bi <= bj) {
                            const int t = rs_base<NFB, W>(s) + bj - bi;
                            f32x4 a = acc[t];
                            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[s], Bh, a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[s], Bl, a, 0, 0, 0);
                            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[s], Bh, a, 0, 0, 0);
                            acc[t] = a;
                        }
                    }
                }
                __syncthreads();                                 // every wave has read the operands: the next chunk may be staged
            }
            // right-hand side / border: four partial sums per feature (one per wave) -> y[16 bi + r] for the owner of row bi.
            // The scratch lives in the panel region, which the elimination only writes after the barriers below.
            float* ysum = Pan;                                   // [4][256]: every lane writes its four features (4 lane .. 4 lane + 3)
            float* bsum = Pan + 4 * 256;                         // [4][256]
            float* ys = Pan + 8 * 256;                           // [FP] y, [FP] b, [8] c / e per wave
            *reinterpret_cast<float4*>(&ysum[W * 256 + 4 * lane]) = make_float4(yreg[0], yreg[1], yreg[2], yreg[3]);
            if constexpr (BORDER) *reinterpret_cast<float4*>(&bsum[W * 256 + 4 * lane]) = make_float4(breg[0], breg[1], breg[2], breg[3]);
            if constexpr (BORDER) { if (lane == 0) { ys[2 * C::FP + 2 * W] = creg; ys[2 * C::FP + 2 * W + 1] = ereg; } }
            yreg[0] = yreg[1] = yreg[2] = yreg[3] = 0.f;
            breg[0] = breg[1] = breg[2] = breg[3] = 0.f;
            creg = ereg = 0.f;
            __syncthreads();
            if (tid < C::FP) {
                ys[tid] = (ysum[tid] + ysum[256 + tid]) + (ysum[2 * 256 + tid] + ysum[3 * 256 + tid]);
                if constexpr (BORDER) ys[C::FP + tid] = (bsum[tid] + bsum[256 + tid]) + (bsum[2 * 256 + tid] + bsum[3 * 256 + tid]);
            }
            __syncthreads();
            constexpr int NTALL = NFB * (NFB + 1) / 2;
            if constexpr (MODE == 1) {                           // this segment's partial system -> global memory, next item
                float* out = partial + it * (int64_t)RS_PARTIAL(NFB, BORDER);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int bi = rs_row<W>(s);
                    if (bi < NFB) {
#pragma unroll
                        for (int bj = bi; bj < NFB; ++bj) {
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg)
                                out[rs_tile_w<NFB>(bi, bj) * 256 + reg * 64 + lane] = acc[rs_base<NFB, W>(s) + bj - bi][reg];
                        }
                    }
                }
                if (tid < C::FP) {
                    out[NTALL * 256 + tid] = ys[tid];
                    if constexpr (BORDER) out[NTALL * 256 + C::FP + tid] = ys[C::FP + tid];
                }
                if constexpr (BORDER) { if (tid < 8) out[NTALL * 256 + 2 * C::FP + tid] = ys[2 * C::FP + tid]; }
                u = un; lo = lon; d = dn;
                __syncthreads();                                 // the scratch is reused by the next item
                continue;
            }
            if constexpr (MODE == 2) {                           // the sum of the row's segments (wmf_launch_combine_segments)
                const float* in = partial + seg_first[it] * (int64_t)RS_PARTIAL(NFB, BORDER);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int bi = rs_row<W>(s);
                    if (bi < NFB) {
#pragma unroll
                        for (int bj = bi; bj < NFB; ++bj) {
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg)
                                acc[rs_base<NFB, W>(s) + bj - bi][reg] = in[rs_tile_w<NFB>(bi, bj) * 256 + reg * 64 + lane];
                        }
                    }
                }
                if (tid < C::FP) {
                    ys[tid] = in[NTALL * 256 + tid];
                    if constexpr (BORDER) ys[C::FP + tid] = in[NTALL * 256 + C::FP + tid];
                }
                if constexpr (BORDER) { if (tid < 8) ys[2 * C::FP + tid] = in[NTALL * 256 + 2 * C::FP + tid]; }
                __syncthreads();
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int bi = rs_row<W>(s);
                if (bi < NFB) {                                  // the q = 0 lanes carry the whole sum (the elimination adds the four q shares)
                    yacc[s] = q == 0 ? ys[16 * bi + r] : 0.f;
                    if constexpr (BORDER) bacc[s] = q == 0 ? ys[C::FP + 16 * bi + r] : 0.f;
                }
            }
            if constexpr (BORDER) {
                const float* ce = ys + 2 * C::FP;
                cacc = q == 0 ? (ce[0] + ce[2]) + (ce[4] + ce[6]) : 0.f;
                eacc = q == 0 ? (ce[1] + ce[3]) + (ce[5] + ce[7]) : 0.f;
            }
            __syncthreads();                                     // the scratch is free: pivot 0 may publish its panels
        } else {
            auto stage = [&](int c) {                                // registers -> LDS buffer c & 1
                float* vb = Vs + (c & 1) * (C::RC * C::LDV);
                const int nrow = min(C::RC, d - c * C::RC);
    #pragma unroll
                for (int i = 0; i < C::PF; ++i)
                    if (pj[i] < C::RC) *reinterpret_cast<float4*>(&vb[pj[i] * C::LDV + 4 * pc[i]]) = pre[i];   // zeros beyond nrow
                if (tid < C::RC) { wsm[(c & 1) * C::RC + tid] = wpre; psm[(c & 1) * C::RC + tid] = (tid < nrow) ? wpre + 1.f : 0.f; }
            };
            auto request = [&](int c) {                              // chunk c of this row, or the next row's first chunk
                if (c < nchunks) load_chunk(lo, d, c * C::RC);
                else if (itn < count) load_chunk(lon, dn, 0);
            };
            if (nchunks > 0) {
                stage(0);
                __syncthreads();
                request(1);
            } else {
                request(0);                                          // nchunks == 0: straight to the next row
            }
            for (int c = 0; c < nchunks; ++c) {
                const int nrow = min(C::RC, d - c * C::RC);
                const float* vb = Vs + (c & 1) * (C::RC * C::LDV);
                const float* wb = wsm + (c & 1) * C::RC;
                const float* pb = psm + (c & 1) * C::RC;
                const int nsteps = WMF_ABL(dbg, 2) ? 0 : (nrow + 3) >> 2;  // dbg: timing ablations (wmf_debug_set_flags)
                for (int ks = 0; ks < nsteps; ++ks) {
                    const float wq = wb[4 * ks + q], pq = pb[4 * ks + q];
                    const float* vrow = vb + (4 * ks + q) * C::LDV + r;
                    float fw[NFB];                                   // w * fragment: the B operands of every tile of column bj
    #pragma unroll
                    for (int fb = 0; fb < NFB; ++fb) fw[fb] = vrow[16 * fb] * wq;
                    float bw = 0.f;
                    if constexpr (BORDER) {
                        const float bf = vb[(4 * ks + q) * C::LDV + 16 * NFB];   // feature f - 1 of this lane's entry (same for all r)
                        bw = bf * wq;
                        cacc += bf * bw;
                        eacc += bf * pq;
                    }
    #pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int bi = rs_row<W>(s);
                        if (bi < NFB) {
                            const float fa = vrow[16 * bi];          // this row's fragment again (A operand): an LDS read is cheaper than a register
                            yacc[s] += fa * pq;
                            if constexpr (BORDER) bacc[s] += fa * bw;
    #pragma unroll
                            for (int bj = bi; bj < NFB; ++bj) {
                                const int t = rs_base<NFB, W>(s) + bj - bi;
                                acc[t] = WMF_MFMA16(fa, fw[bj], acc[t]);
                            }
                        }
                    }
                }
                if (c + 1 < nchunks) {
                    stage(c + 1);
                    __syncthreads();
                    request(c + 2);
                }
            }

        }

        // ---- C: block elimination
        bool ok = true;
        float tbv = 0.f;                                         // BORDER: the solution's last component
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (rs_row<W>(s) < NFB) {
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) if (r == 4 * q + reg) acc[rs_base<NFB, W>(s)][reg] += 1.f;
            }
        }
        // pivot-row work of row slot S (compile time); the caller checks p == rs_row<W>(S)
        auto pivot_row = [&](auto slot_c, int p, float* P1, float* P2, float* wv_out) {
            constexpr int S = decltype(slot_c)::value;
            constexpr int bi = rs_row<W>(S);
            constexpr int b0 = rs_base<NFB, W>(S);
            if constexpr (bi < NFB) {
                f32x4 X = acc[b0];
                gj_inv_sweep<RS_GJ_LDS, true, true>(X, baddr, r, q, ok, std::make_integer_sequence<int, 16>{});   // (pivots above WMF_PIVOT_CAP bounce the row)
                float yp = yacc[S];
                yp = wmf_qsum(yp);
                float wv0 = X[0] * yp, wv1 = X[1] * yp, wv2 = X[2] * yp, wv3 = X[3] * yp;
                wmf_row16_sum4(wv0, wv1, wv2, wv3);              // w_p[4q + reg] on the whole 16-lane row
                if (r == 0) {
                    *reinterpret_cast<float4*>(wv_out + 4 * q) = make_float4(wv0, wv1, wv2, wv3);
                    *reinterpret_cast<float4*>(gs + 16 * bi + 4 * q) = make_float4(wv0, wv1, wv2, wv3);   // start of the backward pass
                }
                if constexpr (BORDER) {                          // the border column rides along like a second right-hand side
                    float bp = bacc[S];
                    bp = wmf_qsum(bp);
                    float wb0 = X[0] * bp, wb1 = X[1] * bp, wb2 = X[2] * bp, wb3 = X[3] * bp;
                    wmf_row16_sum4(wb0, wb1, wb2, wb3);
                    if (r == 0) {
                        *reinterpret_cast<float4*>(wv_out + 16 + 4 * q) = make_float4(wb0, wb1, wb2, wb3);
                        *reinterpret_cast<float4*>(Wball + 16 * bi + 4 * q) = make_float4(wb0, wb1, wb2, wb3);
                    }
                    const float b0v = __shfl(bp, 4 * q), b1v = __shfl(bp, 4 * q + 1), b2v = __shfl(bp, 4 * q + 2), b3v = __shfl(bp, 4 * q + 3);
                    float cc = b0v * wb0 + b1v * wb1 + b2v * wb2 + b3v * wb3;      // this q group's rows of b_p^T w^b_p
                    float ec = b0v * wv0 + b1v * wv1 + b2v * wv2 + b3v * wv3;
                    cc = wmf_qsum(cc);
                    ec = wmf_qsum(ec);
                    if (lane == 0) { bsc[0] += cc; bsc[1] += ec; }   // one pivot owner at a time, barriers in between
                }
                if constexpr (F16A) {
                    // split-f16 tile products (wmf_dw_elim.h, F16T): W'_pj = -X B_pj by two f16 MFMAs; the panels carry the
                    // split operands themselves, one 16-byte word per lane and tile
                    f32x4 nX;
#pragma unroll
                    for (int e = 0; e < 4; ++e) nX[e] = -X[e];
                    const rs_f16x8 xn = rs_split_natural(nX), xh = rs_dup_hi(xn), xl = rs_dup_lo(xn);
                    rs_f16x8* Q1 = reinterpret_cast<rs_f16x8*>(P1);
                    rs_f16x8* Q2 = reinterpret_cast<rs_f16x8*>(P2);
#pragma unroll
                    for (int bj = bi + 1; bj < NFB; ++bj) {
                        const int t = b0 + bj - bi;
                        const rs_f16x8 nb = rs_split_natural(acc[t]);
                        Q1[bj * 64 + lane] = nb;
                        f32x4 n = f32x4{0.f, 0.f, 0.f, 0.f};
                        n = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, nb, n, 0, 0, 0);
                        n = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl, nb, n, 0, 0, 0);
                        Q2[bj * 64 + lane] = rs_split_natural(n);
                        acc[t] = n;                              // W'_pj = -W_pj stays in registers for the backward pass
                    }
                } else {
#pragma unroll
                    for (int bj = bi + 1; bj < NFB; ++bj) {
                        const int t = b0 + bj - bi;
                        f32x4 n = f32x4{0.f, 0.f, 0.f, 0.f};
                        n = WMF_MFMA16(X[0], acc[t][0], n); n = WMF_MFMA16(X[1], acc[t][1], n);
                        n = WMF_MFMA16(X[2], acc[t][2], n); n = WMF_MFMA16(X[3], acc[t][3], n);
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            P1[bj * 320 + (4 * q + reg) * 20 + r] = acc[t][reg];
                            P2[bj * 320 + (4 * q + reg) * 20 + r] = n[reg];
                        }
                        acc[t] = n;                              // W_pj stays in registers for the backward pass
                    }
                }
            }
            (void)p;
        };
        if (!WMF_ABL(dbg, 1)) {
#pragma unroll 1
            for (int p = 0; p < NFB; ++p) {
                float* P1 = Pan;                                 // originals of block row p
                float* P2 = Pan + NFB * (F16A ? 256 : 320);      // W tiles of block row p
                float* wvp = Wv;
                if (p == rs_row<W>(0)) pivot_row(std::integral_constant<int, 0>{}, p, P1, P2, wvp);
                else if (p == rs_row<W>(1)) pivot_row(std::integral_constant<int, 1>{}, p, P1, P2, wvp);
                else if (p == rs_row<W>(2)) pivot_row(std::integral_constant<int, 2>{}, p, P1, P2, wvp);
                else if (p == rs_row<W>(3)) pivot_row(std::integral_constant<int, 3>{}, p, P1, P2, wvp);
                __syncthreads();                                 // panels of row p published
                const float4 w4 = *reinterpret_cast<const float4*>(wvp + 4 * q);
                float4 wb4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (BORDER) wb4 = *reinterpret_cast<const float4*>(wvp + 16 + 4 * q);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    constexpr int dummy = 0; (void)dummy;
                    const int bi = rs_row<W>(s);
                    if (bi < NFB && bi > p) {                    // compile-time row, run-time pivot: a uniform branch per row
                        if constexpr (F16A) {
                            const rs_f16x8 nbi = reinterpret_cast<const rs_f16x8*>(P1)[bi * 64 + lane];
                            float a[4];                          // -B_pi[4q + e][r], rebuilt from its two parts (22 bits: enough for y)
#pragma unroll
                            for (int e = 0; e < 4; ++e) a[e] = -((float)nbi[e] + (float)nbi[4 + e]);
                            yacc[s] += a[0] * w4.x + a[1] * w4.y + a[2] * w4.z + a[3] * w4.w;
                            if constexpr (BORDER) bacc[s] += a[0] * wb4.x + a[1] * wb4.y + a[2] * wb4.z + a[3] * wb4.w;
                            const rs_f16x8 ah = rs_dup_hi(nbi), al = rs_dup_lo(nbi);
#pragma unroll
                            for (int bj = bi; bj < NFB; ++bj) {
                                const int t = rs_base<NFB, W>(s) + bj - bi;
                                const rs_f16x8 nwj = reinterpret_cast<const rs_f16x8*>(P2)[bj * 64 + lane];
                                f32x4 c = acc[t];
                                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, nwj, c, 0, 0, 0);      // B_ij += B_pi^T W'_pj
                                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, nwj, c, 0, 0, 0);
                                acc[t] = c;
                            }
                        } else {
                            float a[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) a[e] = -P1[bi * 320 + (4 * q + e) * 20 + r];
                            yacc[s] += a[0] * w4.x + a[1] * w4.y + a[2] * w4.z + a[3] * w4.w;
                            if constexpr (BORDER) bacc[s] += a[0] * wb4.x + a[1] * wb4.y + a[2] * wb4.z + a[3] * wb4.w;
#pragma unroll
                            for (int bj = bi; bj < NFB; ++bj) {
                                const int t = rs_base<NFB, W>(s) + bj - bi;
#pragma unroll
                                for (int e = 0; e < 4; ++e) acc[t] = WMF_MFMA16(a[e], P2[bj * 320 + (4 * q + e) * 20 + r], acc[t]);
                            }
                        }
                    }
                }
                __syncthreads();                                 // everyone has read the panels: the next pivot may overwrite them
            }
            float tb = 0.f;                                      // BORDER: the last unknown
            if constexpr (BORDER) {
                cacc = wmf_qsum(cacc);
                eacc = wmf_qsum(eacc);
                const float piv = 1.f + cacc - bsc[0];           // identity + c - sum_p b_p^T w^b_p (the last pivot's barrier made bsc final)
                if (!(piv > 1e-20f)) ok = false;
                tb = (eacc - bsc[1]) * __builtin_amdgcn_rcpf(piv);
                tbv = tb;
            }
            if (!ok && lane == 0) flag[0] = 1;
            __syncthreads();
            if constexpr (BORDER) {                              // the backward pass starts from w^y_p - t w^b_p
                for (int c = tid; c < 16 * NFB; c += C::NTHR) gs[c] -= tb * Wball[c];
                __syncthreads();
            }
            // ---- D: rows from the last to the first; gs[16 p ..] holds w_p until row p is finished
#pragma unroll 1
            for (int p = NFB - 1; p >= 0; --p) {
                auto back_row = [&](auto slot_c) {
                    constexpr int S = decltype(slot_c)::value;
                    constexpr int bi = rs_row<W>(S);
                    constexpr int b0 = rs_base<NFB, W>(S);
                    if constexpr (bi < NFB && bi + 1 < NFB) {
                        float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int bj = bi + 1; bj < NFB; ++bj) {
                            const float gj = gs[16 * bj + r];    // g_j[r], final
#pragma unroll
                            for (int reg = 0; reg < 4; ++reg) s4[reg] += (F16A ? -acc[b0 + bj - bi][reg] : acc[b0 + bj - bi][reg]) * gj;   // (F16A: the tile holds -W_pj)
                        }
                        wmf_row16_sum4(s4[0], s4[1], s4[2], s4[3]);
                        if (r == 0) {
                            float* z = gs + 16 * bi + 4 * q;
                            z[0] -= s4[0]; z[1] -= s4[1]; z[2] -= s4[2]; z[3] -= s4[3];
                        }
                    }
                };
                if (p == rs_row<W>(0)) back_row(std::integral_constant<int, 0>{});
                else if (p == rs_row<W>(1)) back_row(std::integral_constant<int, 1>{});
                else if (p == rs_row<W>(2)) back_row(std::integral_constant<int, 2>{});
                else if (p == rs_row<W>(3)) back_row(std::integral_constant<int, 3>{});
                __syncthreads();
            }
        } else {
            __syncthreads();
        }
        const bool notpd = flag[0] != 0;
        if (notpd) {
            if (tid == 0) fb_rows[atomicAdd(fb_count, 1)] = u;   // not positive definite: the LU kernel redoes it
        } else {
            for (int c = tid; c < ld; c += C::NTHR) {
                float v = (c < f) ? gs[c < 16 * NFB ? c : 0] : 0.f;
                if constexpr (BORDER) { if (c == 16 * NFB) v = tbv; else if (c > 16 * NFB) v = 0.f; }
                g[(int64_t)u * ld + c] = v;
            }
        }
        u = un; lo = lon; d = dn;
        __syncthreads();                                         // gs / flag are reused by the next row
    }
}

#ifndef RS_OCC16
#define RS_OCC16 2
#endif
// (The same kernel at NFB = 8 -- k = 128, nine tiles per wave, four workgroups per CU -- was measured against the one-wave-
// per-row LDS-DMA kernel on cfg3's item side: 119.7 ms against 22.5 ms, same rows to 3e-7.  The four-wave split pays a
// workgroup barrier and an LDS round trip per pivot tile and per 32 entries; it is what makes f > 144 fit, not a fast path.)
template <int NFB, bool BORDER, bool F16A, int MODE>
__global__ __launch_bounds__(256, F16A ? RS_OCC16 : 2) void solve_rowsplit_kernel(const int32_t* __restrict__ rows, int64_t count,
                                                                const float* __restrict__ V, const float* __restrict__ biasv,
                                                                const int64_t* __restrict__ indptr,
                                                                const int32_t* __restrict__ indices,
                                                                const float* __restrict__ vals, int f, int ld,
                                                                float* __restrict__ g, int32_t* __restrict__ fb_rows,
                                                                int32_t* __restrict__ fb_count, int dbg,
                                                                const int64_t* __restrict__ seg_lo, const int32_t* __restrict__ seg_d,
                                                                const int32_t* __restrict__ seg_first, float* __restrict__ partial,
                                                                const int32_t* __restrict__ count_dev) {
    if (count_dev) count = *count_dev;                           // (the rows the iteration kernel bounced: the count is on the device)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sm = reinterpret_cast<float*>(smem_raw);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    switch (wave) {
        case 0: rs_body<NFB, 0, BORDER, F16A, MODE>(sm, rows, count, V, biasv, indptr, indices, vals, f, ld, g, fb_rows, fb_count, dbg, seg_lo, seg_d, seg_first, partial); break;
        case 1: rs_body<NFB, 1, BORDER, F16A, MODE>(sm, rows, count, V, biasv, indptr, indices, vals, f, ld, g, fb_rows, fb_count, dbg, seg_lo, seg_d, seg_first, partial); break;
        case 2: rs_body<NFB, 2, BORDER, F16A, MODE>(sm, rows, count, V, biasv, indptr, indices, vals, f, ld, g, fb_rows, fb_count, dbg, seg_lo, seg_d, seg_first, partial); break;
        default: rs_body<NFB, 3, BORDER, F16A, MODE>(sm, rows, count, V, biasv, indptr, indices, vals, f, ld, g, fb_rows, fb_count, dbg, seg_lo, seg_d, seg_first, partial); break;
    }
}

template <int NFB, bool BORDER, bool F16A, int MODE>
static void launch_rowsplit_f(const int32_t* rows, int64_t count, const float* V, const float* biasv, const int64_t* indptr,
                              const int32_t* indices, const float* vals, int f, int ld, float* g, const wmf_plan* pl, hipStream_t st,
                              const int32_t* count_dev = nullptr) {
    using C = RsCfg<NFB, BORDER, F16A>;
    constexpr size_t lds = (size_t)C::TOTAL * 4;
    // ROUND-3 FINDING, ROUND-4 RESOLUTION.  The BORDER kernels with 13 .. 15 blocks (f = 209, 225, 241) gave
    // wrong and run-to-run different right-hand sides y whenever two of their workgroups shared a CU.  Those three are the only
    // instantiations that combine a REAL divergent region in the staging (`lane < 4 NFB` with 4 NFB = 52 / 56 / 60 < 64; at
    // NFB = 16 the test is always true) with heavy spilling (216 .. 376 bytes of scratch, ~400 SGPR spills).  Since round 4 the
    // staging and the store of the y / b partial sums are branch-free -- every lane stores, into operand planes padded to sixteen
    // blocks and into 256-float plane rows -- so every instantiation has the control flow of NFB = 16, which was always exact;
    // the 82 KB LDS guard (one workgroup per CU, 1.5 .. 1.7 x slower at those widths) is gone.  RS_ALONE (lab) brings it back.
#ifdef RS_ALONE
    constexpr bool ALONE = BORDER && NFB >= 13 && NFB <= 15;
#else
    constexpr bool ALONE = false;
#endif
    constexpr size_t lds_launch = ALONE && lds < (size_t)82 * 1024 ? (size_t)82 * 1024 : lds;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)solve_rowsplit_kernel<NFB, BORDER, F16A, MODE>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_launch);
        attr_set = true;
    }
    int64_t grid = 256 * 2 * 2;                                  // two resident workgroups per CU, two rounds
    if (grid > count) grid = count;
    static const char* nm = wmf_kname("solve_rowsplit_kernel<%d, %s, %s, %d>", NFB, BORDER ? "true" : "false",
                                      F16A ? "true" : "false", MODE);
    static const char* nmb = wmf_kname("solve_rowsplit_kernel<%d, %s, %s, %d> [bounced]", NFB, BORDER ? "true" : "false",
                                       F16A ? "true" : "false", MODE);
    WMF_LAUNCH(count_dev ? nmb : nm, (solve_rowsplit_kernel<NFB, BORDER, F16A, MODE>), dim3((unsigned)grid), dim3(C::NTHR), lds_launch, st, rows, count, V,
               biasv, indptr, indices, vals, f, ld, g, pl->fallback_rows, pl->fallback_count,
               wmf_debug_flags, pl->seg_lo, pl->seg_d, pl->seg_first, pl->partial, count_dev);
}

template <int NFB, bool BORDER>
static void launch_rowsplit_nfb(const wmf_plan* pl, const float* V, const float* biasv, const int64_t* indptr,
                                const int32_t* indices, const float* vals, int f, int ld, float* g, hipStream_t st) {
    const int32_t* rows = pl->rows[WMF_BIN_GENERAL];
    const int64_t normal = pl->count[WMF_BIN_GENERAL] - pl->heavy_count;
    // split-f16 accumulation and elimination (debug flag 2097152: the f32 MFMA kernel it replaces; that one does not split rows)
    if (wmf_debug_flags & 2097152) {
        launch_rowsplit_f<NFB, BORDER, false, 0>(rows, pl->count[WMF_BIN_GENERAL], V, biasv, indptr, indices, vals, f, ld, g, pl, st);
        return;
    }
    // ROUND 4: the first iter_count of the normal rows go to the matrix-free iteration kernel (wmf_iter.hip), which hands back what
    // it does not solve as a device-side list (as in wmf_directw.hip); debug flag 268435456: off
    const int64_t n_iter = biasv ? 0 : wmf_iter_rows(pl, f, ld, false);      // (biasv: always folded into vals by wmf_launch_solve)
    if (n_iter > 0)
        (void)wmf_launch_iter(rows, n_iter, V, nullptr, indptr, indices, vals, f, ld, g, pl->iter_bounce_rows, pl->fallback_count + 1,
                              pl->iter_stats, pl->iter_info, st);
    if (normal > n_iter) launch_rowsplit_f<NFB, BORDER, true, 0>(rows + n_iter, normal - n_iter, V, biasv, indptr, indices, vals, f, ld, g, pl, st);
    if (n_iter > 0) launch_rowsplit_f<NFB, BORDER, true, 0>(pl->iter_bounce_rows, n_iter, V, biasv, indptr, indices, vals, f, ld, g, pl, st, pl->fallback_count + 1);
    if (pl->heavy_count > 0) {          // rows with more than WMF_HEAVY_T entries: segments by separate workgroups, then one combine each
        launch_rowsplit_f<NFB, BORDER, true, 1>(rows, pl->seg_total, V, biasv, indptr, indices, vals, f, ld, g, pl, st);
        wmf_launch_combine_segments(pl, RS_PARTIAL(NFB, BORDER), st);
        launch_rowsplit_f<NFB, BORDER, true, 2>(rows + normal, pl->heavy_count, V, biasv, indptr, indices, vals, f, ld, g, pl, st);
    }
}

int64_t wmf_rowsplit_partial_floats(int f) {
    const int border = (f % 16 == 1 && f / 16 >= 10) ? 1 : 0;
    const int64_t nfb = border ? f / 16 : (f + 15) / 16;
    return nfb * (nfb + 1) / 2 * 256 + 16 * nfb + (border ? 16 * nfb + 8 : 0);
}

// 144 < f <= 256, and f = 16 m + 1 up to 257 (k = 16 m with biases: m blocks and a border column)
int wmf_rowsplit_supported(int f) { return f > 144 && (f <= 256 || f == 257); }

int wmf_launch_rowsplit(const wmf_plan* pl, const float* V, const float* biasv, const int64_t* indptr,
                        const int32_t* indices, const float* vals, int f, int ld, float* g, hipStream_t st) {
    if (pl->count[WMF_BIN_GENERAL] <= 0) return 0;
    if (f % 16 == 1 && f / 16 >= 10) {
        switch (f / 16) {
#define C_(N) case N: launch_rowsplit_nfb<N, true>(pl, V, biasv, indptr, indices, vals, f, ld, g, st); break;
            C_(10) C_(11) C_(12) C_(13) C_(14) C_(15) C_(16)
#undef C_
            default: return -1;
        }
        return 0;
    }
    switch ((f + 15) / 16) {
#define C_(N) case N: launch_rowsplit_nfb<N, false>(pl, V, biasv, indptr, indices, vals, f, ld, g, st); break;
        C_(10) C_(11) C_(12) C_(13) C_(14) C_(15) C_(16)
#undef C_
        default: return -1;
    }
    return 0;
}
