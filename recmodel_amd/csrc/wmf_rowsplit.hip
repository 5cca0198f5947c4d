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

template <int NFB, bool BORDER = false>
struct RsCfg {
    static constexpr int NW = 4, NTHR = 256;
    static constexpr int FP = 16 * NFB;
    // row stride = 16 (mod 32): the two k rows of a half wave hit disjoint banks; BORDER needs room for column FP
    static constexpr int LDV = (FP % 32 == 16) ? (BORDER ? FP + 32 : FP) : FP + 16;
    static constexpr int RC = 16;                                 // staged entries per chunk
    static constexpr int PF = (RC * (FP / 4 + (BORDER ? 1 : 0)) + NTHR - 1) / NTHR;  // 16-byte pieces prefetched per thread and chunk
    // LDS carve (floats)
    static constexpr int OFF_VS = 0;                              // [2][RC][LDV] staged factor rows, double buffered
    static constexpr int OFF_W = OFF_VS + 2 * RC * LDV;           // [2][RC] weights
    static constexpr int OFF_P = OFF_W + 2 * RC;                  // [2][RC] w + 1 (0 past the end of the row)
    static constexpr int OFF_PAN = OFF_P + 2 * RC;                 // [2][NFB][16][20]: {originals, W} of the pivot row, by block column
    static constexpr int OFF_WV = OFF_PAN + 2 * NFB * 320;        // [16] w_p of the pivot row (+ spare)
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

template <int NFB, int W, bool BORDER>
__device__ __forceinline__ void rs_body(float* __restrict__ sm, const int32_t* __restrict__ rows, int64_t count,
                                        const float* __restrict__ V, const float* __restrict__ biasv,
                                        const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                        const float* __restrict__ vals, int f, int ld, float* __restrict__ g,
                                        int32_t* __restrict__ fb_rows, int32_t* __restrict__ fb_count, int dbg) {
    using C = RsCfg<NFB, BORDER>;
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
    for (int e = tid; e < 2 * C::RC * C::LDV; e += C::NTHR) Vs[e] = 0.f;   // pad columns [ld, LDV) stay zero for good

    int64_t it = blockIdx.x;
    int u = 0, d = 0;
    int64_t lo = 0;
    if (it < count) {
        u = rows[it]; lo = indptr[u]; d = (int)(indptr[u + 1] - lo);
        load_chunk(lo, d, 0);
    }
    for (; it < count; it += gridDim.x) {
        const int nchunks = (d + C::RC - 1) / C::RC;
        const int64_t itn = it + gridDim.x;
        int un = 0, dn = 0;
        int64_t lon = 0;
        if (itn < count) { un = rows[itn]; lon = indptr[un]; dn = (int)(indptr[un + 1] - lon); }

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
                gj_inv_sweep(X, baddr, r, q, ok, std::make_integer_sequence<int, 16>{});
                float yp = yacc[S];
                yp += __shfl_xor(yp, 16);
                yp += __shfl_xor(yp, 32);
                float wv0 = X[0] * yp, wv1 = X[1] * yp, wv2 = X[2] * yp, wv3 = X[3] * yp;
                wmf_row16_sum4(wv0, wv1, wv2, wv3);              // w_p[4q + reg] on the whole 16-lane row
                if (r == 0) {
                    *reinterpret_cast<float4*>(wv_out + 4 * q) = make_float4(wv0, wv1, wv2, wv3);
                    *reinterpret_cast<float4*>(gs + 16 * bi + 4 * q) = make_float4(wv0, wv1, wv2, wv3);   // start of the backward pass
                }
                if constexpr (BORDER) {                          // the border column rides along like a second right-hand side
                    float bp = bacc[S];
                    bp += __shfl_xor(bp, 16);
                    bp += __shfl_xor(bp, 32);
                    float wb0 = X[0] * bp, wb1 = X[1] * bp, wb2 = X[2] * bp, wb3 = X[3] * bp;
                    wmf_row16_sum4(wb0, wb1, wb2, wb3);
                    if (r == 0) {
                        *reinterpret_cast<float4*>(wv_out + 16 + 4 * q) = make_float4(wb0, wb1, wb2, wb3);
                        *reinterpret_cast<float4*>(Wball + 16 * bi + 4 * q) = make_float4(wb0, wb1, wb2, wb3);
                    }
                    const float b0v = __shfl(bp, 4 * q), b1v = __shfl(bp, 4 * q + 1), b2v = __shfl(bp, 4 * q + 2), b3v = __shfl(bp, 4 * q + 3);
                    float cc = b0v * wb0 + b1v * wb1 + b2v * wb2 + b3v * wb3;      // this q group's rows of b_p^T w^b_p
                    float ec = b0v * wv0 + b1v * wv1 + b2v * wv2 + b3v * wv3;
                    cc += __shfl_xor(cc, 16); cc += __shfl_xor(cc, 32);
                    ec += __shfl_xor(ec, 16); ec += __shfl_xor(ec, 32);
                    if (lane == 0) { bsc[0] += cc; bsc[1] += ec; }   // one pivot owner at a time, barriers in between
                }
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
                    acc[t] = n;                                  // W_pj stays in registers for the backward pass
                }
            }
            (void)p;
        };
        if (!WMF_ABL(dbg, 1)) {
#pragma unroll 1
            for (int p = 0; p < NFB; ++p) {
                float* P1 = Pan;                                 // originals of block row p
                float* P2 = Pan + NFB * 320;                     // W tiles of block row p
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
                __syncthreads();                                 // everyone has read the panels: the next pivot may overwrite them
            }
            float tb = 0.f;                                      // BORDER: the last unknown
            if constexpr (BORDER) {
                cacc += __shfl_xor(cacc, 16); cacc += __shfl_xor(cacc, 32);
                eacc += __shfl_xor(eacc, 16); eacc += __shfl_xor(eacc, 32);
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
                            for (int reg = 0; reg < 4; ++reg) s4[reg] += acc[b0 + bj - bi][reg] * gj;
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

template <int NFB, bool BORDER>
__global__ __launch_bounds__(256, 2) void solve_rowsplit_kernel(const int32_t* __restrict__ rows, int64_t count,
                                                                const float* __restrict__ V, const float* __restrict__ biasv,
                                                                const int64_t* __restrict__ indptr,
                                                                const int32_t* __restrict__ indices,
                                                                const float* __restrict__ vals, int f, int ld,
                                                                float* __restrict__ g, int32_t* __restrict__ fb_rows,
                                                                int32_t* __restrict__ fb_count, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sm = reinterpret_cast<float*>(smem_raw);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    switch (wave) {
        case 0: rs_body<NFB, 0, BORDER>(sm, rows, count, V, biasv, indptr, indices, vals, f, ld, g, fb_rows, fb_count, dbg); break;
        case 1: rs_body<NFB, 1, BORDER>(sm, rows, count, V, biasv, indptr, indices, vals, f, ld, g, fb_rows, fb_count, dbg); break;
        case 2: rs_body<NFB, 2, BORDER>(sm, rows, count, V, biasv, indptr, indices, vals, f, ld, g, fb_rows, fb_count, dbg); break;
        default: rs_body<NFB, 3, BORDER>(sm, rows, count, V, biasv, indptr, indices, vals, f, ld, g, fb_rows, fb_count, dbg); break;
    }
}

template <int NFB, bool BORDER>
static void launch_rowsplit_nfb(const int32_t* rows, int64_t count, const float* V, const float* biasv, const int64_t* indptr,
                                const int32_t* indices, const float* vals, int f, int ld, float* g, int32_t* fb_rows,
                                int32_t* fb_count, hipStream_t st) {
    using C = RsCfg<NFB, BORDER>;
    constexpr size_t lds = (size_t)C::TOTAL * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)solve_rowsplit_kernel<NFB, BORDER>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr_set = true;
    }
    int64_t grid = 256 * 2 * 2;                                  // two resident workgroups per CU, two rounds
    if (grid > count) grid = count;
    static const char* nm = wmf_kname("solve_rowsplit_kernel<%d, %s>", NFB, BORDER ? "true" : "false");
    WMF_LAUNCH(nm, (solve_rowsplit_kernel<NFB, BORDER>), dim3((unsigned)grid), dim3(C::NTHR), lds, st, rows, count, V, biasv,
               indptr, indices, vals, f, ld, g, fb_rows, fb_count, wmf_debug_flags);
}

// 144 < f <= 256, and f = 16 m + 1 up to 257 (k = 16 m with biases: m blocks and a border column)
int wmf_rowsplit_supported(int f) { return f > 144 && (f <= 256 || f == 257); }

int wmf_launch_rowsplit(const int32_t* rows, int64_t count, const float* V, const float* biasv, const int64_t* indptr,
                        const int32_t* indices, const float* vals, int f, int ld, float* g, int32_t* fb_rows,
                        int32_t* fb_count, hipStream_t st) {
    if (count <= 0) return 0;
    if (f % 16 == 1 && f / 16 >= 10) {
        switch (f / 16) {
#define C_(N) case N: launch_rowsplit_nfb<N, true>(rows, count, V, biasv, indptr, indices, vals, f, ld, g, fb_rows, fb_count, st); break;
            C_(10) C_(11) C_(12) C_(13) C_(14) C_(15) C_(16)
#undef C_
            default: return -1;
        }
        return 0;
    }
    switch ((f + 15) / 16) {
#define C_(N) case N: launch_rowsplit_nfb<N, false>(rows, count, V, biasv, indptr, indices, vals, f, ld, g, fb_rows, fb_count, st); break;
        C_(10) C_(11) C_(12) C_(13) C_(14) C_(15) C_(16)
#undef C_
        default: return -1;
    }
    return 0;
}
