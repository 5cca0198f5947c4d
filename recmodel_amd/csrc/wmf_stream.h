// Per-wave streaming of one CSR row's entries into MFMA operand registers (gfx950), shared by the
// one-wave-per-row direct kernels.
//
// Lane (r = l & 15, q = l >> 4).  MFMA k-step t of group g consumes the four entries 4 GS g + 4 t + q
// (q = 0..3 are the four k slots); the A/B operand of "virtual feature block" vb is one feature of row idx.
// FEATURE PERMUTATION: the f x f system is built and solved in a permuted feature order so that the
// gather can use 16-byte loads: lane r takes the pieces V[idx][4 (r + 16 j) .. +3], j < J = NFB / 4, and
// register e of piece j is virtual block 4 j + e (real feature 64 j + 4 r + e); the NFB % 4 remaining
// blocks are dword loads of feature 64 J + 16 rr + r.  One dwordx4 instruction then moves four whole
// 256-byte row segments instead of four 64-byte ones (4x fewer vector-memory instructions; the dword
// form was bound by the address/tag pipeline, not by HBM).  The solve is invariant under the
// permutation; only the final store maps virtual (block, lane) back to the real column (real_col()).
//   * entry metadata travels in BLOCKS of 64 entries: lane l keeps (index, weight, p = weight + 1) of entry
//     64 c + l, two blocks resident; a group picks its four entries out of the block with ds_bpermute.
//     Entries past the row's end are clamped to the last real entry and get weight 0 and p 0, which
//     cancels them in every product (the factor row they read is real, finite data).
//   * factor rows are requested D groups ahead into a ring of D register sets.  The loads are RAW: nothing
//     touches the destination registers until the group is consumed, so the only s_waitcnt for them
//     sits in front of the MFMAs that use them, D - 1 groups of MFMA work after they were issued.
// Every register-array index is a compile-time constant (the group loop is unrolled by D).
#pragma once
#include "wmf_common.h"

template <int NFB, int GS, int D>
struct WmfRowStream {
    static constexpr int J = NFB / 4, R = NFB % 4;  // 16-byte pieces and dword blocks per lane and entry
    static constexpr int EPG = 4 * GS;            // entries per group
    static constexpr int GPB = 64 / EPG;          // groups per 64-entry block
    float fr[D][GS][NFB];                         // factor-row operands
    float w[D][GS];                               // weight of the entry of (k-step t, slot q)
    float p[D][GS];                               // w + 1 for real entries, 0 for padding
    int idxB[2];
    float wB[2], pB[2];
    int idxM[GS];                                 // metadata of the next group to be requested (fetched one call early,
    float wM[GS], pM[GS];                         // so the ds_bpermute latency is off the address path)

    // block c of the row (lo, d): lane l <- entry 64 c + l
    __device__ __forceinline__ void load_block(int c, int64_t lo, int d, const int32_t* __restrict__ indices,
                                               const float* __restrict__ vals, int lane, int slot) {
        const int j = 64 * c + lane;
        const float mask = j < d ? 1.f : 0.f;
        const int64_t e = d > 0 ? lo + max(min(j, d - 1), 0) : 0;   // d == 0 (a row without entries): entry 0, weight 0
        const int idx = indices[e];
        const float wv = vals[e];
        if (slot == 0) { idxB[0] = idx; wB[0] = wv * mask; pB[0] = (wv + 1.f) * mask; }
        else           { idxB[1] = idx; wB[1] = wv * mask; pB[1] = (wv + 1.f) * mask; }
    }

    // pick the entries of group g out of their (resident) block
    __device__ __forceinline__ void fetch_meta(int g, int q) {
        const int c = g / GPB;
        const int idx_c = (c & 1) ? idxB[1] : idxB[0];
        const float w_c = (c & 1) ? wB[1] : wB[0];
        const float p_c = (c & 1) ? pB[1] : pB[0];
#pragma unroll
        for (int t = 0; t < GS; ++t) {
            const int src = ((EPG * g + 4 * t + q) & 63) << 2;                       // ds_bpermute byte address
            idxM[t] = __builtin_amdgcn_ds_bpermute(src, idx_c);
            wM[t] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, w_c)));
            pM[t] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, p_c)));
        }
    }

    // request the factor rows of group g into ring slot S (compile time); fetch_meta(g) must have run,
    // and groups are requested in increasing order, so the metadata of g + 1 is fetched on the way out.
    // side != NULL (split layout, wmf_internal.h; R >= 1): rows of V are ld floats of packed body, and the last dword block
    // comes from the pairs: lane r = 0 gets the row's last feature, lanes r >= 1 its bias.
    template <int S>
    __device__ __forceinline__ void load_group(int g, const float* __restrict__ V, int ld, int r, int q,
                                               const float* __restrict__ side = nullptr) {
        const int nch = ld >> 2;
#pragma unroll
        for (int t = 0; t < GS; ++t) {
            w[S][t] = wM[t];
            p[S][t] = pM[t];
            const float* vrow = V + (int64_t)idxM[t] * ld;
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const int c = (R == 0 && j == J - 1) ? min(r + 16 * j, nch - 1) : r + 16 * j;   // only the last piece can run past the row
                const float4 v = reinterpret_cast<const float4*>(vrow)[c];
                fr[S][t][4 * j] = v.x; fr[S][t][4 * j + 1] = v.y; fr[S][t][4 * j + 2] = v.z; fr[S][t][4 * j + 3] = v.w;
            }
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const int col = 64 * J + 16 * rr + r;
                if (rr == R - 1 && side) fr[S][t][4 * J + rr] = side[2 * (int64_t)idxM[t] + min(r, 1)];
                else fr[S][t][4 * J + rr] = vrow[rr == R - 1 ? min(col, ld - 1) : col];
            }
        }
        fetch_meta(g + 1, q);
    }

    // consumer side: zero the features that lie beyond the row (their loads were clamped)
    template <int S>
    __device__ __forceinline__ void mask_tail(int t, int ld, int r) {
        if constexpr (R == 0) {
            const float m = (4 * (r + 16 * (J - 1)) < ld) ? 1.f : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) fr[S][t][4 * (J - 1) + e] *= m;
        } else {
            fr[S][t][NFB - 1] *= (64 * J + 16 * (R - 1) + r < ld) ? 1.f : 0.f;
        }
    }

    // real column of virtual block vb on lane r
    static __device__ __forceinline__ int real_col(int vb, int r) {
        return vb < 4 * J ? 64 * (vb >> 2) + 4 * r + (vb & 3) : 64 * J + 16 * (vb - 4 * J) + r;
    }
};
