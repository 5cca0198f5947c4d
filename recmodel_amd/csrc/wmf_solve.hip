// Per-row normal-equation solve of one ALS half step, in whitened coordinates (gfx950).
//
// Reference loop body: RecModel/wmf_model.py:220-239 (no bias) / :337-350 (bias):
//     A_u = G + U^T diag(w) U,  b_u = (w + 1)^T U,  x_u = solve(A_u, b_u),  U = Y[idx_u]
// With G = L L^T and V = Y~ L^-T (wmf_dense.hip) this is  x_u = L^-T g_u  with
//     g_u = (I + V_u^T D V_u)^-1 V_u^T p,           p = w + 1, D = diag(w)      ("direct", f x f)
//         = V_u^T (I + D S)^-1 p,  S = V_u V_u^T                                 ("low rank", d x d)
// The two forms are algebraically identical (push-through identity); the second costs O(d^2 f)
// instead of O(d f^2 + f^3) and is used for rows with few stored entries, which is most of them.
// Only V is gathered: one 4f-byte row per stored entry, the same bytes the reference's Y[idx] reads.
//
// Kernels
//   solve_low<NCH, 1>   d <= 16, one wave per row: S by f32 MFMA straight from global loads,
//   solve_low<NCH, 2>   d <= 32   Gauss-Jordan on (I + D S) across the wave, g by DPP row sums.
//   solve_general<NFB>  any d, any sign of w: f x f system in LDS, LU with partial pivoting
//                       (the reference's np.linalg.solve is LAPACK gesv = the same algorithm).
#include "wmf_common.h"
#include "wmf_internal.h"

#include <utility>

__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// One Gauss-Jordan step on pivot K (compile time, so register indices and the DPP control are
// static).  Row j is spread over the 4 lanes (r, q = 0..3); column K lives in lanes q = (K & 15) >> 2,
// register (K >> 4) * 4 + (K & 3).
template <int K, int NSETS>
__device__ __forceinline__ void gj_step(float (&m)[NSETS][NSETS * 4], float (&p)[NSETS], const int (&baddr)[4], int r) {
    constexpr int ks = K >> 4, kk = K & 15, kq = kk >> 2, kreg = ks * 4 + (kk & 3);
    const float piv = readlane_f(m[ks][kreg], kk + 16 * kq);
    const float inv = __builtin_amdgcn_rcpf(piv);
    const float pkv = readlane_f(p[ks], kk);
    // multiplier -M[j][K] / piv per row (M[j][K] sits in lane (r, kq): one ds_bpermute with a precomputed
    // address); for the pivot row itself -(piv - 1) / piv, which turns row - f * row into row / piv (the
    // normalised pivot row) without a select per element.  The other set goes first: it must read the
    // pivot row before that row is rewritten.
    if constexpr (NSETS == 2) {
        constexpr int so = 1 - ks;
        const float fo = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(baddr[kq], __builtin_bit_cast(int, m[so][kreg])));
        const float nf = -fo * inv;
#pragma unroll
        for (int c4 = 0; c4 < NSETS; ++c4)       // m[so][c] += nf * (lane kk of this 16-lane row of m[ks][c])
            fmac_bcast4<kk>(m[so][4 * c4], m[so][4 * c4 + 1], m[so][4 * c4 + 2], m[so][4 * c4 + 3], m[ks][4 * c4],
                            m[ks][4 * c4 + 1], m[ks][4 * c4 + 2], m[ks][4 * c4 + 3], nf);
        p[so] = fmaf(nf, pkv, p[so]);
    }
    {
        const float fk = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(baddr[kq], __builtin_bit_cast(int, m[ks][kreg])));
        // nf = (e_K - f) / piv with e_K = [r == kk] as a constant lane mask shifted into place next to its use (three
        // instructions where  r == kk ? piv - 1 : f  and the product take four; see gj_inv_step_lean, wmf_common.h)
        float eK;
        unsigned long long tmp;
        asm volatile("s_lshl_b64 %1, %2, %3\n\tv_cndmask_b32 %0, 0, 1.0, %1" : "=v"(eK), "=&s"(tmp) : "s"(0x0001000100010001ull), "n"(kk) : "scc");
        const float nf = (eK - fk) * inv;
#pragma unroll
        for (int c4 = 0; c4 < NSETS; ++c4)
            fmac_bcast4_self<kk>(m[ks][4 * c4], m[ks][4 * c4 + 1], m[ks][4 * c4 + 2], m[ks][4 * c4 + 3], nf);
        p[ks] = fmaf(nf, pkv, p[ks]);
    }
}

// Steps run in groups of four under one scalar branch.  A step past the row's last entry is a no-op
// (its pivot row is an identity row, its column is zero everywhere else), so at most three are wasted.
template <int G, int NSETS>
__device__ __forceinline__ void gj_group(float (&m)[NSETS][NSETS * 4], float (&p)[NSETS], int d, const int (&baddr)[4],
                                         int r) {
    if (4 * G < d) {                                // d is wave-uniform (SGPR)
        gj_step<4 * G, NSETS>(m, p, baddr, r);
        gj_step<4 * G + 1, NSETS>(m, p, baddr, r);
        gj_step<4 * G + 2, NSETS>(m, p, baddr, r);
        gj_step<4 * G + 3, NSETS>(m, p, baddr, r);
    }
}
template <int NSETS, int... Gs>
__device__ __forceinline__ void gj_sweep(float (&m)[NSETS][NSETS * 4], float (&p)[NSETS], int d, const int (&baddr)[4],
                                         int r, std::integer_sequence<int, Gs...>) {
    (gj_group<Gs, NSETS>(m, p, d, baddr, r), ...);
}

// ------------------------------------------------------------------------------ low-degree rows
// Lane (r = l & 15, q = l >> 4).  Stored entry j of the row is handled by the four lanes with
// r == j (set A) or r == j - 16 (set B, NSETS == 2).  Each lane loads 16-byte pieces
// V[idx][16 t + 4 q .. +3]; element e of piece t is the MFMA operand of step (t, e) for k slot q.
// Because S = V_u V_u^T, the same register is the A and the B operand.
#ifndef WMF_LOW_ABLATE
#define WMF_LOW_ABLATE 0
#endif
// d <= 16 rows are latency bound: six waves per SIMD (<= 80 registers, no spills up to NCH = 9) measured 15 % faster
// than the five the compiler settles on at k = 128; the d <= 32 kernel needs its 100 registers.
// X6 (the row's features are whole 32-feature chunks -- ld a multiple of 32 -- or whole chunks and ONE more 16-byte piece:
// f = 32 c + 1 with ld = f + 3, e.g. k = 128 with biases, row-major or in the split layout): the S tiles by split-f16
// products.  A lane then takes the pieces 8 c + 2 q, 8 c + 2 q + 1 of its rows (the eight k of chunk c's 16x16x32 MFMA
// operand) instead of 4 t + q; every gathered value, scaled by 2^8, is split into two f16 parts x = hi + lo (22 bits) and
// each tile gets three f16 MFMAs per 32 features (lo.hi, hi.lo, hi.hi) instead of eight f32 ones; the odd last piece is one
// f32 MFMA step.  An f16 MFMA holds the SIMD's pipe for 8 of its 16 cycles, an f32 one for all 32 -- and these kernels are
// bound by exactly that pipe (fetching 20 % fewer bytes per row left solve_low<9, 1> at cfg3 where it was, 8.6 ms).
// Whitened rows have |v| <= 1 (V^T V <= I), so the scaled values stay far below the f16 range, and their low parts (2^-11
// of the value) stay normal f16 numbers down to |v| ~ 5e-4; the error of a tile entry is that of its f32 accumulation
// (the rounds before this used three bf16 parts and six MFMAs: exact split, twice the MFMA time).
typedef _Float16 low_f16x8 __attribute__((ext_vector_type(8)));
#define WMF_LOW_SC 256.f
template <int E> __device__ __forceinline__ float low_elem(const float4& a, const float4& b) {
    return E == 0 ? a.x : E == 1 ? a.y : E == 2 ? a.z : E == 3 ? a.w : E == 4 ? b.x : E == 5 ? b.y : E == 6 ? b.z : b.w;
}
// the two f16 parts of the eight values of pieces (a, b), scaled
__device__ __forceinline__ void low_split8(const float4& a, const float4& b, low_f16x8& hi, low_f16x8& lo) {
    // (the scaling by a power of two is exact: split the scaled values)
    const wmf_u32x4 s0 = wmf_split4(a.x * WMF_LOW_SC, a.y * WMF_LOW_SC, a.z * WMF_LOW_SC, a.w * WMF_LOW_SC);
    const wmf_u32x4 s1 = wmf_split4(b.x * WMF_LOW_SC, b.y * WMF_LOW_SC, b.z * WMF_LOW_SC, b.w * WMF_LOW_SC);
    hi = __builtin_bit_cast(low_f16x8, wmf_u32x4{s0[0], s0[1], s1[0], s1[1]});
    lo = __builtin_bit_cast(low_f16x8, wmf_u32x4{s0[2], s0[3], s1[2], s1[3]});
}
// bstride = 3: the rolled coordinates of wmf_row_transform mode 3 (include/wmf_hip.h) -- the border feature is biasv[0] for every
// row and the 32 bits of an entry's bias are the last mantissa bits of its body features 8 j, 8 j + 1 (j < 16).  In the X6 piece
// order lane (r, q) holds piece 2 j = 8 c + 2 q of entry r for c = 0 .. 3, i.e. j = 4 c + q: eight of the bits; the entry's four lanes
// (r, r + 16, r + 32, r + 48) OR theirs together.  pieces[2 c] = piece 8 c + 2 q.
template <class P>
__device__ __forceinline__ float wmf_lsb_bias(const P& pieces, int q) {
    unsigned b = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const unsigned two = (__builtin_bit_cast(unsigned, pieces[2 * c].x) & 1u) | ((__builtin_bit_cast(unsigned, pieces[2 * c].y) & 1u) << 1);
        b |= two << (8 * c + 2 * q);
    }
    b |= (unsigned)__shfl_xor((int)b, 16);
    b |= (unsigned)__shfl_xor((int)b, 32);
    return __builtin_bit_cast(float, b);
}

template <int NCH, int NSETS, bool BLK, bool X6 = false>
__global__ __launch_bounds__(256, (NSETS == 1 && NCH <= 9) ? 6 : ((X6 && NCH <= 4) ? 5 : 1)) void solve_low_kernel(const int32_t* __restrict__ rows, int64_t count,
                                                        const float* __restrict__ V, const float* __restrict__ biasv,
                                                        const int64_t* __restrict__ indptr,
                                                        const int32_t* __restrict__ indices,
                                                        const float* __restrict__ vals, int ld, int last1, float* __restrict__ g,
                                                        int32_t* __restrict__ fb_rows, int32_t* __restrict__ fb_count, int bstride) {
    // biasv / bstride: bstride = 1: biasv[idx] is the fixed side's bias of row idx.  bstride = 2: the SPLIT LAYOUT
    // (wmf_internal.h): V holds packed body rows of ld - 4 floats, biasv the pairs {last feature, bias} of the rows -- the
    // last 16-byte piece of a gathered row is then made of the pair, which every lane of the entry loads (8 bytes, L2)
    const int lane = threadIdx.x & 63;
    // wave-uniform values are forced into SGPRs: hipcc cannot see that threadIdx.x >> 6 is uniform, and
    // would otherwise predicate every `k < d` step with exec masks and register copies
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
    if (wid >= count) return;                       // whole wave exits together
    const int u = __builtin_amdgcn_readfirstlane(rows[wid]);
    const int64_t lo_v = indptr[u];
    const int64_t lo = ((int64_t)__builtin_amdgcn_readfirstlane((int)(lo_v >> 32)) << 32) |
                       (uint32_t)__builtin_amdgcn_readfirstlane((int)lo_v);
    const int d = __builtin_amdgcn_readfirstlane((int)(indptr[u + 1] - lo));
    const int r = lane & 15, q = lane >> 4;
    const int nch = ld >> 2;
    const int vch = bstride >= 2 ? nch - 1 : nch;   // 16-byte pieces of a stored row of V
    const float cb = bstride == 3 ? biasv[0] : 0.f; // (3) every row's border value

    // All loads are unconditional and clamped to a stored entry (under a per-lane select hipcc sinks the load
    // into a branch and waits for each in turn).  Slots j >= d read the factor row of entry 0 -- real, finite
    // data -- with weight 0 and p 0: their row of I + D S is an identity row and their c_j is exactly 0, so
    // they drop out of every product without masking the 16 x NCH gathered values.  Only the last piece can
    // run past the row's ld features; it is clamped and multiplied by 0.
    float w[NSETS], p[NSETS];
    float4 x[NSETS][NCH];
    bool neg = false;
    const float4* Vq = reinterpret_cast<const float4*>(V) + q;      // this lane's pieces are q, q+4, q+8, ...
    constexpr bool TAIL = X6 && (NCH & 1);          // X6: whole chunks and one more piece, the row's last (lane q = 0 owns it)
    const int last_c = TAIL ? nch - 1 : min(4 * (NCH - 1) + q, nch - 1);
    const float last_m = TAIL ? (q == 0 ? 1.f : 0.f) : ((4 * (NCH - 1) + q < nch) ? 1.f : 0.f);
#pragma unroll
    for (int s = 0; s < NSETS; ++s) {
        const int j = r + 16 * s;
        const float am = j < d ? 1.f : 0.f;
        const int64_t e = j < d ? lo + j : 0;       // entry 0 exists: the launcher skips matrices without entries
        const int idx = indices[e];
        float wj = vals[e];
        if (biasv && bstride == 1) wj -= biasv[idx];
        const float4* vrow = Vq + (int64_t)idx * vch;
        if constexpr (X6) {                          // pieces 8 c + 2 q + h of the whole chunks
#pragma unroll
            for (int t = 0; t < NCH - (TAIL ? 1 : 0); ++t) x[s][t] = (vrow - q)[8 * (t >> 1) + 2 * q + (t & 1)];
        } else {
#pragma unroll
            for (int t = 0; t < NCH - 1; ++t) x[s][t] = vrow[4 * t];
        }
        if constexpr (!X6 || TAIL) {
            float4 v;
            if (bstride == 2) {                          // split layout: f = 4 (nch - 1) + 1, the last piece is { feature f - 1, bias, 0, 0 }
                const float2 sd = reinterpret_cast<const float2*>(biasv)[idx];
                v = make_float4(sd.x, sd.y, 0.f, 0.f);
            } else if (bstride == 3) {                   // ... and neither half of it is fetched
                if constexpr (X6 && NCH == 9) v = make_float4(cb, wmf_lsb_bias(x[s], q), 0.f, 0.f);
                else v = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                v = (vrow - q)[last_c];
            }
            const float pad_m = (bstride > 1) ? 0.f : last_m;
            if (bstride > 1) wj -= v.y;
            x[s][NCH - 1] = make_float4(v.x * last_m, v.y * pad_m, v.z * last_m, v.w * last_m);
        }
        wj *= am;
        if (!(wj >= 0.f)) neg = true;               // negative or NaN weight: needs pivoting
        w[s] = wj;
        p[s] = wj + am;
    }
    if (__any(neg)) {                               // wave-uniform: bounce the row to the LU kernel
        if (lane == 0) fb_rows[atomicAdd(fb_count, 1)] = u;
        return;
    }

    int baddr[4];                                   // ds_bpermute byte address of lane (r, kq), kq = 0..3
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) baddr[kq] = (r + 16 * kq) * 4;
    // ---- S blocks by MFMA: Sb[s][c][reg] = S[row r + 16 s][col 16 c + 4 q + reg].  acc layout is D[4q + reg][r];
    //      S symmetric, so tile(a = x[c], b = x[s]) = S[set c row 4q+reg][set s row r] = S[set s row r][set c row 4q+reg].
    float Sb[NSETS][NSETS][4];
#if WMF_LOW_ABLATE & 1                                 // timing experiments only (tools/kernel_lab.py): no S tiles
#pragma unroll
    for (int s = 0; s < NSETS; ++s)
#pragma unroll
        for (int c = 0; c < NSETS; ++c)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) Sb[s][c][reg] = 1e-3f * (x[s][reg % NCH].x + x[c][(reg + 1) % NCH].y + x[s][NCH - 1].z);
#else
    {
        // one accumulator per tile; the NSETS^2 tiles are interleaved so that dependent MFMAs are NSETS^2 - 1
        // (NSETS = 1: 0, hence two chains there) instructions apart
        f32x4 acc[NSETS][NSETS], acc1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NSETS; ++s)
#pragma unroll
            for (int c = 0; c < NSETS; ++c) acc[s][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if constexpr (X6) {
            constexpr int NKC = (NCH - (TAIL ? 1 : 0)) / 2;
            static_assert(NKC >= 1 && 2 * NKC + (TAIL ? 1 : 0) == NCH, "X6: whole 32-feature chunks, at most one more piece");
#pragma unroll
            for (int cc = 0; cc < NKC; ++cc) {
                low_f16x8 hi[NSETS], lo[NSETS];
#pragma unroll
                for (int s = 0; s < NSETS; ++s) low_split8(x[s][2 * cc], x[s][2 * cc + 1], hi[s], lo[s]);
#pragma unroll
                for (int s = 0; s < NSETS; ++s)
#pragma unroll
                    for (int c = s; c < NSETS; ++c) {             // tile (s, c): A from set c, B from set s, as the f32 path
                        // (NSETS == 1: two accumulation chains, even and odd chunks)
                        f32x4 a = (NSETS == 1 && (cc & 1)) ? acc1 : acc[s][c];
                        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(lo[c], hi[s], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi[c], lo[s], a, 0, 0, 0);
                        a = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi[c], hi[s], a, 0, 0, 0);
                        if (NSETS == 1 && (cc & 1)) acc1 = a; else acc[s][c] = a;
                    }
            }
            constexpr float UNSC = 1.f / (WMF_LOW_SC * WMF_LOW_SC);
#pragma unroll
            for (int s = 0; s < NSETS; ++s)
#pragma unroll
                for (int c = s; c < NSETS; ++c) acc[s][c] *= UNSC;
            acc1 *= UNSC;
            if constexpr (TAIL) {                    // the row's last piece holds one feature (and, bias models, the bias: masked)
#pragma unroll
                for (int s = 0; s < NSETS; ++s)
#pragma unroll
                    for (int c = s; c < NSETS; ++c) acc[s][c] = WMF_MFMA16(x[c][NCH - 1].x, x[s][NCH - 1].x, acc[s][c]);
            }
        }
#pragma unroll
        for (int t = 0; t < (X6 ? 0 : NCH); ++t) {
            // last1: the row's last piece holds ONE feature (f = 4 m + 1, e.g. k = 128 with the bias column) and three padding
            // zeros: three of its four k-steps would multiply zeros
            const bool tail_only = (t == NCH - 1) && last1;
            if constexpr (NSETS == 1) {
                acc[0][0] = WMF_MFMA16(x[0][t].x, x[0][t].x, acc[0][0]);
                if (!tail_only) {
                    acc1 = WMF_MFMA16(x[0][t].y, x[0][t].y, acc1);
                    acc[0][0] = WMF_MFMA16(x[0][t].z, x[0][t].z, acc[0][0]);
                    acc1 = WMF_MFMA16(x[0][t].w, x[0][t].w, acc1);
                }
            } else {
                // S_BA = S_AB^T is not accumulated: it is one tile transpose (4 MFMAs against the identity) below
#define WMF_S4(E)                                                                   \
    _Pragma("unroll") for (int s = 0; s < NSETS; ++s)                               \
        _Pragma("unroll") for (int c = s; c < NSETS; ++c) acc[s][c] = WMF_MFMA16(x[c][t].E, x[s][t].E, acc[s][c]);
                WMF_S4(x) WMF_S4(y) WMF_S4(z) WMF_S4(w)       // (no tail skip here: the branch cost this kernel more than the 9 MFMAs)
#undef WMF_S4
            }
        }
        if constexpr (NSETS == 2) {
            // MFMA(RD(X), RD(Y)) = RD(Y X^T) for row-distributed tiles: with Y = I this is the transpose, exactly
            f32x4 tr = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) tr = WMF_MFMA16(acc[0][1][e], (r == 4 * q + e) ? 1.f : 0.f, tr);
            acc[1][0] = tr;
        }
#pragma unroll
        for (int s = 0; s < NSETS; ++s)
#pragma unroll
            for (int c = 0; c < NSETS; ++c)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg) {
                    if constexpr (NSETS == 1) Sb[s][c][reg] = acc[s][c][reg] + acc1[reg];
                    else Sb[s][c][reg] = acc[s][c][reg];
                }
    }
#endif
#if WMF_LOW_ABLATE & 2                                 // no solve: c = p + (something that keeps S alive)
#pragma unroll
    for (int s = 0; s < NSETS; ++s)
#pragma unroll
        for (int c = 0; c < NSETS; ++c) p[s] += w[s] * (Sb[s][c][0] + Sb[s][c][1] + Sb[s][c][2] + Sb[s][c][3]);
    if constexpr (false) {
#else
    if constexpr (BLK && NSETS == 2) {
#endif
        // ---- symmetric form and 2 x 2 block elimination with 16 x 16 tiles:
        //   c = E y,   (I + E S E) y = E^-1 p,   E = diag(sqrt(w))        (I + D S = E (I + E S E) E^-1)
        //   P = [[A, U], [B, C]],  B = U^T:  X = A^-1 (tile Gauss-Jordan), T = B X and S' = C - T B^T by MFMA on the
        //   row-distributed registers (MFMA(RD(X), RD(Y)) = RD(Y X^T)), then two tile solves for the vectors.
        // (Round 3: this replaces  c = p - E y', (I + E S E) y' = E S p,  whose last step subtracts two numbers of the size of
        // w to get a c of order one -- exact enough at w ~ 10, a relative error of 6e-8 w / c beyond: wrong rows for
        // confidence weights of 1e4 and more, tests/test_gpu_parity.py::test_weight_range_of_the_class_surface.)
        // A weight of zero -- a stored zero, which still contributes p = 1 (wmf_model.py:232,239), or a slot past the row's end
        // (p = 0) -- has no E^-1: it is taken as 1e-30, whose row of P is an identity row to 1e-15: y = 1e15 p there and
        // c = E y = p, exactly what the row of I + D S says.
        const float wc0 = fmaxf(w[0], 1e-30f), wc1 = fmaxf(w[1], 1e-30f);
        const float e0 = __builtin_amdgcn_sqrtf(wc0), e1 = __builtin_amdgcn_sqrtf(wc1);   // raw v_sqrt_f32 (1 ulp): E only has to satisfy E^2 ~ D
        const int caddr[4] = {(4 * q + 0) * 4, (4 * q + 1) * 4, (4 * q + 2) * 4, (4 * q + 3) * 4};   // lane of entry 4q + reg (q = 0 row)
        auto col4 = [&](float v, float (&out)[4]) {             // out[reg] = value of row 4q + reg
#pragma unroll
            for (int reg = 0; reg < 4; ++reg)
                out[reg] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(caddr[reg], __builtin_bit_cast(int, v)));
        };
        auto qsum = [&](float v) { return wmf_qsum(v); };
        float eA[4], eB[4];
        col4(e0, eA);
        col4(e1, eB);
        f32x4 PA, PU, PB, PC;
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float dg = (r == 4 * q + reg) ? 1.f : 0.f;
            PA[reg] = dg + e0 * Sb[0][0][reg] * eA[reg];
            PU[reg] = e0 * Sb[0][1][reg] * eB[reg];
            PB[reg] = e1 * Sb[1][0][reg] * eA[reg];
            PC[reg] = dg + e1 * Sb[1][1][reg] * eB[reg];
        }
        const float tA = p[0] * __builtin_amdgcn_rsqf(wc0);      // E^-1 p
        float tB = p[1] * __builtin_amdgcn_rsqf(wc1);
        bool okb = true;
        f32x4 X = PA;
        gj_inv_sweep<true, false>(X, baddr, r, q, okb, std::make_integer_sequence<int, 16>{});   // w >= 0: I + E S_AA E is SPD, pivots >= 1
        f32x4 T = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) T = WMF_MFMA16(X[e], PB[e], T);               // RD(B X)
        f32x4 SC = PC;
#pragma unroll
        for (int e = 0; e < 4; ++e) SC = WMF_MFMA16(-T[e], PB[e], SC);            // RD(C - T B^T), symmetric
        float col[4];
        col4(tA, col);
        tB -= qsum(T[0] * col[0] + T[1] * col[1] + T[2] * col[2] + T[3] * col[3]);
        float mS[1][4] = {{SC[0], SC[1], SC[2], SC[3]}};
        float yv[1] = {tB};
        gj_sweep<1>(mS, yv, d - 16, baddr, r, std::make_integer_sequence<int, 4>{});   // rows past d - 16 are identity rows
        const float yB = yv[0];
        col4(yB, col);
        const float z = tA - qsum(PU[0] * col[0] + PU[1] * col[1] + PU[2] * col[2] + PU[3] * col[3]);
        col4(z, col);
        const float yA = qsum(X[0] * col[0] + X[1] * col[1] + X[2] * col[2] + X[3] * col[3]);
        p[0] = e0 * yA;
        p[1] = e1 * yB;
        if (!okb) p[0] = __builtin_nanf("");                     // caught by the finite check below
    } else {
        // ---- M = I + D S row-distributed: m[s][c*4 + reg] = M[row r + 16 s][col 16 c + 4 q + reg]; Gauss-Jordan
        //      without pivoting (row-scaled SPD when w >= 0: pivots >= 1).  After the sweep p = M^-1 p = c.
        float m[NSETS][NSETS * 4];
#pragma unroll
        for (int s = 0; s < NSETS; ++s)
#pragma unroll
            for (int c = 0; c < NSETS; ++c)
#pragma unroll
                for (int reg = 0; reg < 4; ++reg)
                    m[s][c * 4 + reg] = w[s] * Sb[s][c][reg] + ((s == c && r == 4 * q + reg) ? 1.f : 0.f);
#if !(WMF_LOW_ABLATE & 2)
        gj_sweep<NSETS>(m, p, d, baddr, r, std::make_integer_sequence<int, 4 * NSETS>{});
#endif
    }
    // With w >= 0 every pivot is >= 1 in exact arithmetic, so the sweep cannot break down; a NaN/Inf
    // in the inputs is what is left to catch, and it survives into c.
    bool bad = false;
#pragma unroll
    for (int s = 0; s < NSETS; ++s) if (!(fabsf(p[s]) < 3.0e38f)) bad = true;
    if (__any(bad)) {                               // wave-uniform: let the pivoted LU kernel report on the row
        if (lane == 0) fb_rows[atomicAdd(fb_count, 1)] = u;
        return;
    }

    // ---- g = V_u^T c: scale own pieces by c_j, sum over the 16 lanes of the DPP row, store.
    float4* grow = reinterpret_cast<float4*>(g + (int64_t)u * ld);
#pragma unroll
    for (int t = 0; t < NCH; ++t) {
        float4 y = make_float4(p[0] * x[0][t].x, p[0] * x[0][t].y, p[0] * x[0][t].z, p[0] * x[0][t].w);
        if constexpr (NSETS == 2) {
            y.x += p[1] * x[1][t].x; y.y += p[1] * x[1][t].y; y.z += p[1] * x[1][t].z; y.w += p[1] * x[1][t].w;
        }
#if !(WMF_LOW_ABLATE & 4)
        wmf_row16_sum4_scatter(y.x, y.y, y.z, y.w);     // y.x = the sum of component r >> 2, on the lanes r = 0, 4, 8, 12 among others
#endif
        // the piece this lane holds in slot t (X6 with a last odd piece: lane q = 0 owns it)
        const int c = X6 ? ((TAIL && t == NCH - 1) ? (q == 0 ? nch - 1 : nch) : 8 * (t >> 1) + 2 * q + (t & 1)) : 4 * t + q;
        if ((r & 3) == 0 && c < nch) reinterpret_cast<float*>(grow)[4 * c + (r >> 2)] = y.x;
    }
}

// ------------------------------------------------------------------------------ rows with d <= 8
// TWO rows per wave: row A's entries in lanes r = 0..7 of every 16-lane group, row B's in r = 8..15.  The one
// 16 x 16 S tile then holds S_A and S_B on its diagonal 8 x 8 blocks (the cross blocks are dropped when M = I + D S is
// formed), one Gauss-Jordan sweep solves both systems (steps 0..7 belong to A, 8..15 to B), and the DPP sums of
// g = V_u^T c stop after three stages: half the MFMAs and two thirds of the VALU work per row of solve_low<NCH, 1>.
// A third of cfg3's users (Poisson(10) degrees) and most users of a power-law data set are such rows.
template <int NCH, bool X6 = false>
__global__ __launch_bounds__(256, NCH <= 9 ? 6 : 1) void solve_pair_kernel(const int32_t* __restrict__ rows, int64_t count,
                                                                          const float* __restrict__ V, const float* __restrict__ biasv,
                                                                          const int64_t* __restrict__ indptr,
                                                                          const int32_t* __restrict__ indices,
                                                                          const float* __restrict__ vals, int ld, int last1, float* __restrict__ g,
                                                                          int32_t* __restrict__ fb_rows, int32_t* __restrict__ fb_count, int bstride) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t pid = (int64_t)blockIdx.x * 4 + wave;
    if (2 * pid >= count) return;                    // whole wave exits together
    const bool hasB = 2 * pid + 1 < count;
    const int uA = __builtin_amdgcn_readfirstlane(rows[2 * pid]);
    const int uB = __builtin_amdgcn_readfirstlane(hasB ? rows[2 * pid + 1] : rows[2 * pid]);
    const int64_t loA = indptr[uA], loB = indptr[uB];
    const int dA = __builtin_amdgcn_readfirstlane((int)(indptr[uA + 1] - loA));
    const int dB = hasB ? __builtin_amdgcn_readfirstlane((int)(indptr[uB + 1] - loB)) : 0;
    const int r = lane & 15, q = lane >> 4;
    const int nch = ld >> 2;
    const bool second = r >= 8;                      // this lane's entry belongs to row B
    const int j = r & 7;
    const int my_d = second ? dB : dA;
    const float am = j < my_d ? 1.f : 0.f;
    const int64_t e = j < my_d ? (second ? loB : loA) + j : 0;     // entry 0 exists: the launcher skips matrices without entries
    const int idx = indices[e];
    float wj = vals[e];
    if (biasv && bstride == 1) wj -= biasv[idx];
    float4 x[NCH];
    const float4* Vq = reinterpret_cast<const float4*>(V) + q;
    constexpr bool TAIL = X6 && (NCH & 1);           // X6 (solve_low_kernel): whole 32-feature chunks and at most one more piece
    const int last_c = TAIL ? nch - 1 : min(4 * (NCH - 1) + q, nch - 1);
    const float last_m = TAIL ? (q == 0 ? 1.f : 0.f) : ((4 * (NCH - 1) + q < nch) ? 1.f : 0.f);
    const float4* vrow = Vq + (int64_t)idx * (bstride >= 2 ? nch - 1 : nch);
    const float cb = bstride == 3 ? biasv[0] : 0.f;  // (3) every row's border value (solve_low_kernel)
    if constexpr (X6) {
#pragma unroll
        for (int t = 0; t < NCH - (TAIL ? 1 : 0); ++t) x[t] = (vrow - q)[8 * (t >> 1) + 2 * q + (t & 1)];
    } else {
#pragma unroll
        for (int t = 0; t < NCH - 1; ++t) x[t] = vrow[4 * t];
    }
    if constexpr (!X6 || TAIL) {
        float4 v;
        if (bstride == 2) {                              // split layout: the last piece from the {last feature, bias} pair (solve_low_kernel)
            const float2 sd = reinterpret_cast<const float2*>(biasv)[idx];
            v = make_float4(sd.x, sd.y, 0.f, 0.f);
        } else if (bstride == 3) {
            if constexpr (X6 && NCH == 9) v = make_float4(cb, wmf_lsb_bias(x, q), 0.f, 0.f);
            else v = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            v = (vrow - q)[last_c];
        }
        const float pad_m = (bstride > 1) ? 0.f : last_m;
        if (bstride > 1) wj -= v.y;
        x[NCH - 1] = make_float4(v.x * last_m, v.y * pad_m, v.z * last_m, v.w * last_m);
    }
    wj *= am;
    const bool neg = !(wj >= 0.f);                   // negative or NaN weight: needs pivoting
    float w[1] = {wj}, p[1] = {wj + am};
    auto bounce = [&]() {                            // both rows go to the pivoted LU kernel
        if (lane == 0) {
            const int at = atomicAdd(fb_count, hasB ? 2 : 1);
            fb_rows[at] = uA;
            if (hasB) fb_rows[at + 1] = uB;
        }
    };
    if (__any(neg)) { bounce(); return; }

    int baddr[4];
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) baddr[kq] = (r + 16 * kq) * 4;
    f32x4 a0 = f32x4{0.f, 0.f, 0.f, 0.f}, a1 = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (X6) {                              // split-f16 products, two chains (even and odd chunks)
        constexpr int NKC = (NCH - (TAIL ? 1 : 0)) / 2;
        static_assert(NKC >= 1 && 2 * NKC + (TAIL ? 1 : 0) == NCH, "X6: whole 32-feature chunks, at most one more piece");
#pragma unroll
        for (int cc = 0; cc < NKC; ++cc) {
            low_f16x8 hi, lo;
            low_split8(x[2 * cc], x[2 * cc + 1], hi, lo);
            f32x4 a = (cc & 1) ? a1 : a0;
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(lo, hi, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, lo, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_f16(hi, hi, a, 0, 0, 0);
            if (cc & 1) a1 = a; else a0 = a;
        }
        constexpr float UNSC = 1.f / (WMF_LOW_SC * WMF_LOW_SC);
        a0 *= UNSC; a1 *= UNSC;
        if constexpr (TAIL) a0 = WMF_MFMA16(x[NCH - 1].x, x[NCH - 1].x, a0);
    }
#pragma unroll
    for (int t = 0; t < (X6 ? 0 : NCH); ++t) {
        a0 = WMF_MFMA16(x[t].x, x[t].x, a0);
        if (!((t == NCH - 1) && last1)) {              // (a last piece of one feature and three padding zeros: solve_low_kernel)
            a1 = WMF_MFMA16(x[t].y, x[t].y, a1);
            a0 = WMF_MFMA16(x[t].z, x[t].z, a0);
            a1 = WMF_MFMA16(x[t].w, x[t].w, a1);
        }
    }
    // M = I + D S restricted to the two diagonal 8 x 8 blocks: this lane's columns are 4q + reg, i.e. block q >> 1
    const float keep = (second == (q >= 2)) ? 1.f : 0.f;
    float m[1][4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) m[0][reg] = w[0] * keep * (a0[reg] + a1[reg]) + ((r == 4 * q + reg) ? 1.f : 0.f);
    if (0 < dA) { gj_step<0, 1>(m, p, baddr, r); gj_step<1, 1>(m, p, baddr, r); gj_step<2, 1>(m, p, baddr, r); gj_step<3, 1>(m, p, baddr, r); }
    if (4 < dA) { gj_step<4, 1>(m, p, baddr, r); gj_step<5, 1>(m, p, baddr, r); gj_step<6, 1>(m, p, baddr, r); gj_step<7, 1>(m, p, baddr, r); }
    if (0 < dB) { gj_step<8, 1>(m, p, baddr, r); gj_step<9, 1>(m, p, baddr, r); gj_step<10, 1>(m, p, baddr, r); gj_step<11, 1>(m, p, baddr, r); }
    if (4 < dB) { gj_step<12, 1>(m, p, baddr, r); gj_step<13, 1>(m, p, baddr, r); gj_step<14, 1>(m, p, baddr, r); gj_step<15, 1>(m, p, baddr, r); }
    if (__any(!(fabsf(p[0]) < 3.0e38f))) { bounce(); return; }

    // ---- g = V_u^T c per row: sums over the 8 lanes of a half, lanes r = 0 and r = 8 store
    float4* grow = reinterpret_cast<float4*>(g + (int64_t)(second ? uB : uA) * ld);
    const bool writer = j == 0 && (!second || hasB);
#pragma unroll
    for (int t = 0; t < NCH; ++t) {
        float4 y = make_float4(p[0] * x[t].x, p[0] * x[t].y, p[0] * x[t].z, p[0] * x[t].w);
        wmf_row8_sum4(y.x, y.y, y.z, y.w);
        const int c = X6 ? ((TAIL && t == NCH - 1) ? (q == 0 ? nch - 1 : nch) : 8 * (t >> 1) + 2 * q + (t & 1)) : 4 * t + q;
        if (writer && c < nch) grow[c] = y;
    }
}

// ---------------------------------------------------------------------------------- general rows
// One 256-thread workgroup per row.  Thread (ty, tx) = (tid >> 4, tid & 15) accumulates the
// NFB x NFB register block B[ty + 16 i][tx + 16 j] of  B = I + V_u^T D V_u  over the row's
// entries, staged RC at a time through LDS; then LU with partial pivoting in LDS.
template <int NFB>
__global__ __launch_bounds__(256) void solve_general_kernel(const int32_t* __restrict__ rows, int64_t count,
                                                            const int32_t* __restrict__ count_ptr,
                                                            const float* __restrict__ V, const float* __restrict__ biasv,
                                                            const int64_t* __restrict__ indptr,
                                                            const int32_t* __restrict__ indices,
                                                            const float* __restrict__ vals, int f, int ld,
                                                            float* __restrict__ g, int32_t* __restrict__ fail_count, int bstride) {
    // (split layout, bstride = 2: V is the packed body, biasv the {last feature, bias} pairs; column f of the staged rows then
    // holds the bias, and only the leading f x f block and the first f entries of the right-hand side are ever used)
    constexpr int FP = 16 * NFB;
    constexpr int LDV = FP + 4;          // staging row stride (floats), keeps 16-byte alignment
    constexpr int LDB = FP + 1;          // odd: conflict-free column walks
    constexpr int RC = 16;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* Vs = reinterpret_cast<float*>(smem_raw);            // [RC][LDV]
    float* ws = Vs + RC * LDV;                                  // [RC] weights
    float* rhs = ws + RC;                                       // [FP]
    float* red = rhs + FP;                                      // [8] reduction scratch
    float* B = red + 8;                                         // [FP][LDB]
    const int tid = threadIdx.x, ty = tid >> 4, tx = tid & 15;
    const int nch = ld >> 2;
    const int64_t total = count_ptr ? (int64_t)*count_ptr : count;

    for (int64_t it = blockIdx.x; it < total; it += gridDim.x) {
        const int u = rows[it];
        const int64_t lo = indptr[u];
        const int d = (int)(indptr[u + 1] - lo);
        float acc[NFB][NFB];
#pragma unroll
        for (int i = 0; i < NFB; ++i)
#pragma unroll
            for (int j = 0; j < NFB; ++j) acc[i][j] = 0.f;
        float racc = 0.f;
        __syncthreads();                                        // previous row's LDS use is over
        for (int e = tid; e < RC * LDV; e += 256) Vs[e] = 0.f;  // zero incl. columns [ld, FP)
        for (int base = 0; base < d; base += RC) {
            const int nrow = min(RC, d - base);
            __syncthreads();
            // stage nrow gathered rows (16-byte pieces) and their weights
            for (int e = tid; e < nrow * nch; e += 256) {
                const int j = e / nch, c = e % nch;
                const int idx = indices[lo + base + j];
                float4 v;
                if (bstride >= 2 && c == nch - 1) {
                    const float2 sd = reinterpret_cast<const float2*>(biasv)[idx];
                    v = make_float4(sd.x, sd.y, 0.f, 0.f);
                } else {
                    v = reinterpret_cast<const float4*>(V + (int64_t)idx * (bstride >= 2 ? ld - 4 : ld))[c];
                }
                *reinterpret_cast<float4*>(&Vs[j * LDV + 4 * c]) = v;
            }
            if (tid < nrow) {
                const int idx = indices[lo + base + tid];
                float wj = vals[lo + base + tid];
                if (biasv) wj -= biasv[bstride >= 2 ? 2 * (int64_t)idx + 1 : (int64_t)idx];
                ws[tid] = wj;
            }
            __syncthreads();
            for (int j = 0; j < nrow; ++j) {
                const float wj = ws[j];
                float va[NFB], vb[NFB];
#pragma unroll
                for (int i = 0; i < NFB; ++i) { va[i] = wj * Vs[j * LDV + ty + 16 * i]; vb[i] = Vs[j * LDV + tx + 16 * i]; }
#pragma unroll
                for (int i = 0; i < NFB; ++i)
#pragma unroll
                    for (int jj = 0; jj < NFB; ++jj) acc[i][jj] += va[i] * vb[jj];
                if (tid < FP) racc += (wj + 1.f) * Vs[j * LDV + tid];
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NFB; ++i)
#pragma unroll
            for (int j = 0; j < NFB; ++j) {
                const int a = ty + 16 * i, b = tx + 16 * j;
                B[a * LDB + b] = acc[i][j] + (a == b ? 1.f : 0.f);
            }
        if (tid < FP) rhs[tid] = racc;
        __syncthreads();

        // ---- LU with partial pivoting on the leading f x f block (LAPACK gesv order).
        bool singular = false;
        for (int k = 0; k < f; ++k) {
            // arg max |B[i][k]|, i >= k  (f <= 144 < 256: one candidate per thread)
            float best = -1.f; int bi = k;
            if (tid >= k && tid < f) { best = fabsf(B[tid * LDB + k]); bi = tid; }
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
            if (!(best > 1e-30f)) { singular = true; break; }   // uniform: every thread reads the same red[]
            if (bi != k) {                                      // swap rows k and bi (columns spread over threads)
                for (int c = tid; c < f; c += 256) { const float a = B[k * LDB + c]; B[k * LDB + c] = B[bi * LDB + c]; B[bi * LDB + c] = a; }
                if (tid == 0) { const float a = rhs[k]; rhs[k] = rhs[bi]; rhs[bi] = a; }
            }
            __syncthreads();
            const float inv = 1.f / B[k * LDB + k];
            const float rk = rhs[k];
            // eliminate below: thread (ty, tx) covers rows k+1+ty+16a, cols k+1+tx+16b
            for (int i = k + 1 + ty; i < f; i += 16) {
                const float l = B[i * LDB + k] * inv;
                for (int c = k + 1 + tx; c < f; c += 16) B[i * LDB + c] -= l * B[k * LDB + c];
                if (tx == 0) rhs[i] -= l * rk;
            }
            __syncthreads();
        }
        if (singular) {
            if (tid == 0) atomicAdd(fail_count, 1);
            for (int c = tid; c < ld; c += 256) g[(int64_t)u * ld + c] = 0.f;
            continue;
        }
        // ---- back substitution (column oriented)
        for (int k = f - 1; k >= 0; --k) {
            const float xk = rhs[k] / B[k * LDB + k];
            __syncthreads();
            if (tid < k) rhs[tid] -= B[tid * LDB + k] * xk;
            if (tid == k) rhs[k] = xk;
            __syncthreads();
        }
        for (int c = tid; c < ld; c += 256) g[(int64_t)u * ld + c] = (c < f) ? rhs[c] : 0.f;
    }
}

// ----------------------------------------------------------------------------------------- spmm
// g[u] = sum_j values[j] * V[indices[j]]  -- one wave per row, lane owns 16-byte pieces.
__global__ __launch_bounds__(256) void spmm_kernel(const float* __restrict__ V, const int64_t* __restrict__ indptr,
                                                   const int32_t* __restrict__ indices, const float* __restrict__ vals,
                                                   int64_t n, int ld, float* __restrict__ g) {
    const int lane = threadIdx.x & 63;
    const int nch = ld >> 2;
    for (int64_t u = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); u < n; u += (int64_t)gridDim.x * 4) {
        const int64_t lo = indptr[u], hi = indptr[u + 1];
        for (int c = lane; c < nch; c += 64) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int64_t j = lo; j < hi; ++j) {
                const float v = vals[j];
                const float4 y = reinterpret_cast<const float4*>(V + (int64_t)indices[j] * ld)[c];
                a.x += v * y.x; a.y += v * y.y; a.z += v * y.z; a.w += v * y.w;
            }
            reinterpret_cast<float4*>(g + u * (int64_t)ld)[c] = a;
        }
    }
}

int wmf_launch_spmm(const float* V, const int64_t* indptr, const int32_t* indices, const float* values, int64_t n,
                    int ld, float* g, hipStream_t st) {
    if (n <= 0) return 0;
    int64_t grid = (n + 3) / 4;
    if (grid > 8192) grid = 8192;
    WmfProfScope ps("spmm_kernel", st);
    hipLaunchKernelGGL(spmm_kernel, dim3((unsigned)grid), dim3(256), 0, st, V, indptr, indices, values, n, ld, g);
    return 0;
}

// ----------------------------------------------------------------------- bias-adjusted weights
// w_eff[j] = values[j] - bias[indices[j]]   (RecModel/wmf_model.py:343), one streaming pass per half step,
// so that the row kernels never chain a dependent gather behind their index loads.
__global__ __launch_bounds__(256) void bias_adjust_kernel(const float* __restrict__ vals, const int32_t* __restrict__ indices,
                                                          const float* __restrict__ biasv, int64_t nnz, int64_t n4,
                                                          float* __restrict__ w_eff) {
    // n4 quads of entries go four per thread and iteration (16-byte streams, four independent bias gathers in
    // flight); the launcher passes n4 = 0 when the caller's arrays are not 16-byte aligned (views into a larger CSR)
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n4; j += (int64_t)gridDim.x * 256) {
        const int4 ix = reinterpret_cast<const int4*>(indices)[j];
        const float4 v = reinterpret_cast<const float4*>(vals)[j];
        const float b0 = biasv[ix.x], b1 = biasv[ix.y], b2 = biasv[ix.z], b3 = biasv[ix.w];
        reinterpret_cast<float4*>(w_eff)[j] = make_float4(v.x - b0, v.y - b1, v.z - b2, v.w - b3);
    }
    for (int64_t j = 4 * n4 + (int64_t)blockIdx.x * 256 + threadIdx.x; j < nnz; j += (int64_t)gridDim.x * 256)
        w_eff[j] = vals[j] - biasv[indices[j]];
}

void wmf_launch_bias_adjust(const float* vals, const int32_t* indices, const float* biasv, int64_t nnz, float* w_eff,
                            hipStream_t st) {
    if (nnz <= 0) return;
    int64_t grid = (nnz + 255) / 256;
    if (grid > 8192) grid = 8192;
    WmfProfScope ps("bias_adjust_kernel", st);
    const bool aligned = ((reinterpret_cast<uintptr_t>(vals) | reinterpret_cast<uintptr_t>(indices)) & 15) == 0;
    hipLaunchKernelGGL(bias_adjust_kernel, dim3((unsigned)grid), dim3(256), 0, st, vals, indices, biasv, nnz,
                       aligned ? (nnz >> 2) : (int64_t)0, w_eff);
}

// ------------------------------------------------------------------------------------- launchers
template <int NCH>
static void launch_low(const wmf_plan* pl, const float* V, const float* biasv, int bstride, const int64_t* indptr,
                       const int32_t* indices, const float* vals, int ld, int last1, float* g, hipStream_t st) {
    const int64_t c0 = pl->count[WMF_BIN_LOW16], c1 = pl->count[WMF_BIN_LOW32];
    // rows with at most 8 entries come first in the bin and go two per wave (solve_pair_kernel)
    const int64_t c8 = (wmf_debug_flags & 2048) ? 0 : pl->count8;
    // split-f16 S tiles (X6, solve_low_kernel) where the row is whole 32-feature chunks, or those and one more piece
    // (debug flag 524288: f32 MFMAs everywhere)
    constexpr bool X6_OK = (NCH % 2 == 0) || (NCH >= 3);
    const bool x6 = X6_OK && !(wmf_debug_flags & 524288) && ((NCH % 2 == 0) ? (ld % 32 == 0) : (last1 && ld == 16 * (NCH - 1) + 4));   // (last1: f = ld - 3, one feature in the last piece)
    if (bstride == 3 && !(x6 && NCH == 9)) bstride = 2;        // (the bias is rebuilt from the row in the X6 piece order only: elsewhere the pairs are read)
#define WMF_LOW_LAUNCH(KERNEL, NAME, ROWS, COUNT, GRID)                                                                   \
    do {                                                                                                                  \
        static const char* nm_ = NAME;                                                                                     \
        WmfProfScope ps_(nm_, st);                                                                                         \
        hipLaunchKernelGGL(KERNEL, dim3((unsigned)(GRID)), dim3(256), 0, st, ROWS, COUNT, V, biasv, indptr, indices, vals,  \
                           ld, last1, g, pl->fallback_rows, pl->fallback_count, bstride);                                   \
    } while (0)
    if (c8 > 0) {
        if constexpr (X6_OK) {
            if (x6) WMF_LOW_LAUNCH((solve_pair_kernel<NCH, true>), wmf_kname("solve_pair_kernel<%d, true>", NCH), pl->rows[WMF_BIN_LOW16], c8, ((c8 + 1) / 2 + 3) / 4);
        }
        if (!x6) WMF_LOW_LAUNCH((solve_pair_kernel<NCH, false>), wmf_kname("solve_pair_kernel<%d, false>", NCH), pl->rows[WMF_BIN_LOW16], c8, ((c8 + 1) / 2 + 3) / 4);
    }
    if (c0 - c8 > 0) {
        if constexpr (X6_OK) {
            if (x6) WMF_LOW_LAUNCH((solve_low_kernel<NCH, 1, false, true>), wmf_kname("solve_low_kernel<%d, 1, false, true>", NCH), pl->rows[WMF_BIN_LOW16] + c8, c0 - c8, (c0 - c8 + 3) / 4);
        }
        if (!x6) WMF_LOW_LAUNCH((solve_low_kernel<NCH, 1, false, false>), wmf_kname("solve_low_kernel<%d, 1, false, false>", NCH), pl->rows[WMF_BIN_LOW16] + c8, c0 - c8, (c0 - c8 + 3) / 4);
    }
    if (c1 > 0) {
        if (wmf_debug_flags & 64) {     // plain 32 x 32 Gauss-Jordan, kept for A/B timing
            WMF_LOW_LAUNCH((solve_low_kernel<NCH, 2, false, false>), wmf_kname("solve_low_kernel<%d, 2, false, false>", NCH), pl->rows[WMF_BIN_LOW32], c1, (c1 + 3) / 4);
        } else {
            if constexpr (X6_OK) {
                if (x6) WMF_LOW_LAUNCH((solve_low_kernel<NCH, 2, true, true>), wmf_kname("solve_low_kernel<%d, 2, true, true>", NCH), pl->rows[WMF_BIN_LOW32], c1, (c1 + 3) / 4);
            }
            if (!x6) WMF_LOW_LAUNCH((solve_low_kernel<NCH, 2, true, false>), wmf_kname("solve_low_kernel<%d, 2, true, false>", NCH), pl->rows[WMF_BIN_LOW32], c1, (c1 + 3) / 4);
        }
    }
#undef WMF_LOW_LAUNCH
}

template <int NFB>
static void launch_general(const int32_t* rows, int64_t count, const int32_t* count_ptr, int grid, const float* V,
                           const float* biasv, int bstride, const int64_t* indptr, const int32_t* indices, const float* vals, int f,
                           int ld, float* g, int32_t* fail_count, hipStream_t st) {
    constexpr int FP = 16 * NFB;
    constexpr size_t lds = ((size_t)16 * (FP + 4) + 16 + FP + 8 + (size_t)FP * (FP + 1)) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)solve_general_kernel<NFB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr_set = true;
    }
    static const char* nm = wmf_kname("solve_general_kernel<%d>", NFB);
    WMF_LAUNCH(nm, (solve_general_kernel<NFB>), dim3(grid), dim3(256), lds, st, rows, count, count_ptr, V, biasv,
               indptr, indices, vals, f, ld, g, fail_count, bstride);
}

static int dispatch_general(const int32_t* rows, int64_t count, const int32_t* count_ptr, int grid, const float* V,
                            const float* biasv, int bstride, const int64_t* indptr, const int32_t* indices, const float* vals, int f,
                            int ld, float* g, int32_t* fail_count, hipStream_t st) {
    switch ((f + 15) / 16) {
#define C(N) case N: launch_general<N>(rows, count, count_ptr, grid, V, biasv, bstride, indptr, indices, vals, f, ld, g, fail_count, st); break;
        C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9)
#undef C
        default: return -1;
    }
    return 0;
}

// ---- split rows: the partial systems of a row's segments summed into the row's first slot.  One thread per float4 of the
// system and row, the segments added one after the other (fixed order: results do not depend on the launch), so that a row
// of a million entries -- 500 segments, tens of megabytes of partial systems -- is combined by hundreds of workgroups and
// not by the one that eliminates it.
__global__ __launch_bounds__(256) void combine_segments_kernel(float* __restrict__ partial, const int32_t* __restrict__ seg_first,
                                                               int64_t heavy_count, int64_t pf4) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= pf4) return;
    for (int64_t h = blockIdx.y; h < heavy_count; h += gridDim.y) {
        const int64_t s0 = seg_first[h], s1 = seg_first[h + 1];
        float4* base = reinterpret_cast<float4*>(partial) + c;
        float4 a = base[s0 * pf4];
        int64_t s = s0 + 1;
        for (; s + 4 <= s1; s += 4) {                            // four loads in flight
            const float4 b0 = base[s * pf4], b1 = base[(s + 1) * pf4], b2 = base[(s + 2) * pf4], b3 = base[(s + 3) * pf4];
            a.x += b0.x; a.y += b0.y; a.z += b0.z; a.w += b0.w;
            a.x += b1.x; a.y += b1.y; a.z += b1.z; a.w += b1.w;
            a.x += b2.x; a.y += b2.y; a.z += b2.z; a.w += b2.w;
            a.x += b3.x; a.y += b3.y; a.z += b3.z; a.w += b3.w;
        }
        for (; s < s1; ++s) { const float4 b = base[s * pf4]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        base[s0 * pf4] = a;
    }
}

void wmf_launch_combine_segments(const wmf_plan* pl, int64_t partial_floats, hipStream_t st) {
    if (pl->heavy_count <= 0) return;
    const int64_t pf4 = partial_floats / 4;                      // (every partial layout is a multiple of four floats)
    const unsigned gy = (unsigned)(pl->heavy_count < 16384 ? pl->heavy_count : 16384);
    WMF_LAUNCH("combine_segments_kernel", combine_segments_kernel, dim3((unsigned)((pf4 + 255) / 256), gy), dim3(256), 0, st,
               pl->partial, pl->seg_first, pl->heavy_count, pf4);
}

int wmf_launch_solve(const wmf_plan* pl, const float* V, const float* biasv, const int64_t* indptr,
                     const int32_t* indices, const float* vals, int f, int ld, float* g, int32_t* fail_count,
                     hipStream_t st) {
    if (hipMemsetAsync(pl->fallback_count, 0, 3 * sizeof(int32_t), st) != hipSuccess) return -2;   // [0] pivoted fallback, [1] rows handed back by wmf_iter.hip, [2] its stage 1's hand-on list
    const int64_t nnz = pl->nnz[0] + pl->nnz[1] + pl->nnz[2] + pl->nnz[3];
    if (nnz == 0)                                                      // nothing stored: every row solves to zero
        return hipMemsetAsync(g, 0, (size_t)pl->n * ld * sizeof(float), st) == hipSuccess ? 0 : -2;
    int bstride = 1;
    const float* side = nullptr;
    if (biasv && wmf_split_layout(f, ld)) {                            // split layout: V is the packed body, biasv the pairs
        side = biasv;
        bstride = (pl->rolled && f == 129) ? 3 : 2;                     // (3: wmf_solve_rows_ex(WMF_SOLVE_ROLLED): the low-row kernels rebuild the bias from the row)
    } else if (biasv) {                                                // other widths: fold the biases into the weights once
        if (!pl->w_eff) return -3;                                     // (plan latched the split layout, this call is not in it)
        wmf_launch_bias_adjust(vals, indices, biasv, nnz, pl->w_eff, st);   // (w_eff: allocated by wmf_plan_create(bias = 1))
        vals = pl->w_eff;
        biasv = nullptr;
    }
    switch ((ld + 15) / 16) {
#define C(N) case N: launch_low<N>(pl, V, biasv, bstride, indptr, indices, vals, ld, (f % 4 == 1 && ld == f + 3 && (ld / 4) % 4 == 1) ? 1 : 0, g, st); break;   /* the lanes' last slot holds that one piece only */
        C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16) C(17)
#undef C
        default: return -1;
    }
    const bool general_ok = f <= 144;
    if (pl->count[WMF_BIN_MFMA] > 0 && (wmf_debug_flags & 33554432)) {
        // debug flag 33554432 (accuracy experiments): every row of this bin through the pivoted float32 LU kernel
        if (dispatch_general(pl->rows[WMF_BIN_MFMA], pl->count[WMF_BIN_MFMA], nullptr, 256, V, biasv, bstride, indptr, indices, vals, f, ld,
                             g, fail_count, st)) return -1;
    } else if (pl->count[WMF_BIN_MFMA] > 0) {
        // one wave per row with the whole system in MFMA accumulator registers (wmf_directw.hip, wmf_directl.hip)
        if (wmf_launch_directw(pl, V, side, indptr, indices, vals, f, ld, g, st)) return -1;
    }
    if (pl->count[WMF_BIN_GENERAL] > 0) {
        // f > 144: rows with more than 32 entries go to the workgroup-per-row kernel (wmf_wide.hip)
        if (!wmf_wide_supported(f)) return -1;
        // f <= 256: four waves per row, tiles owned by block row (wmf_rowsplit.hip); f = 257 .. 272, or debug flag
        // 1024: the run-time-indexed eight-wave kernel (wmf_wide.hip)
        if (wmf_rowsplit_supported(f) && !(wmf_debug_flags & 1024)) {
            if (wmf_launch_rowsplit(pl, V, biasv, indptr, indices, vals, f, ld, g, st)) return -1;
        } else {
            // (f = 258 .. 272: no split rows; the iteration kernel first, as in wmf_directw.hip / wmf_rowsplit.hip)
            const int32_t* rows = pl->rows[WMF_BIN_GENERAL];
            const int64_t all = pl->count[WMF_BIN_GENERAL];
            const int64_t n_iter = biasv ? 0 : wmf_iter_rows(pl, f, ld, false);
            if (n_iter > 0 && wmf_launch_iter(rows, n_iter, V, nullptr, indptr, indices, vals, f, ld, g, pl->iter_bounce_rows,
                                              pl->fallback_count + 1, pl->iter_stats, pl->iter_info, st)) return -1;
            if (all > n_iter && wmf_launch_wide(rows + n_iter, all - n_iter, V, biasv, indptr, indices, vals, f, ld, g,
                                                pl->fallback_rows, pl->fallback_count, st)) return -1;
            if (n_iter > 0 && wmf_launch_wide(pl->iter_bounce_rows, n_iter, V, biasv, indptr, indices, vals, f, ld, g,
                                              pl->fallback_rows, pl->fallback_count, st, pl->fallback_count + 1)) return -1;
        }
    }
    {
        // rows bounced by the other kernels (negative weights / not positive definite); count is on the device
        if (general_ok) {
            if (dispatch_general(pl->fallback_rows, 0, pl->fallback_count, 256, V, biasv, bstride, indptr, indices, vals, f, ld, g,
                                 fail_count, st)) return -1;
        } else {
            if (wmf_launch_wide_lu(pl->fallback_rows, pl->fallback_count, V, biasv, indptr, indices, vals, f, ld, g, fail_count,
                                   pl->wide_ws, st)) return -1;
        }
    }
    return 0;
}
