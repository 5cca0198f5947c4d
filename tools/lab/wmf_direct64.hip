// Heavy rows when the factor width fits one wave (f <= 64): ONE WAVE PER ROW, no workgroup barriers.
//
//   g_u = (I + V_u^T D V_u)^-1 V_u^T p          (RecModel/wmf_model.py:233-239 in whitened coordinates)
//
//   A. the wave streams the row's entries straight from HBM into MFMA operand registers: lane
//      (r = l & 15, q = l >> 4) loads V[idx_{4s+q}][16 fb + r] for k-step s (the Gramian kernel's
//      pattern), one 16-entry group ahead; all NFB(NFB+1)/2 upper tiles of V^T D V accumulate in
//      this wave's registers, rhs = V^T p rides along on the VALU.
//   B. tiles (+ I) go transposed into the lower triangle of an LDS image Bm[FP][FP+4].
//   C. left-looking blocked Cholesky, 16 columns per panel: the panel is first updated with all
//      previous panels by MFMA (operands and C tiles from LDS), then factored in registers with one
//      lane per matrix row (pivot column broadcast by v_readlane, so the rows below the panel get
//      their triangular solve for free); the forward substitution is fused into the same sweep.
//   D. backward substitution with one lane per unknown (row reads of L are conflict free).
// Many independent waves per CU overlap one row's MFMA phase with another row's VALU phase.
#include "wmf_common.h"
#include "wmf_internal.h"
#include "wmf_stream.h"

#include <type_traits>

__device__ __forceinline__ float rl64(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

template <int NFB>
__global__ __launch_bounds__(64, 3) void solve_direct64_kernel(const int32_t* __restrict__ rows, int64_t count,
                                                               const float* __restrict__ V, const float* __restrict__ biasv,
                                                               const int64_t* __restrict__ indptr,
                                                               const int32_t* __restrict__ indices,
                                                               const float* __restrict__ vals, int f, int ld,
                                                               float* __restrict__ g, int32_t* __restrict__ fb_rows,
                                                               int32_t* __restrict__ fb_count, int dbg) {
    constexpr int FP = 16 * NFB;
    constexpr int LDB = FP + 4;
    constexpr int NT = NFB * (NFB + 1) / 2;
    __shared__ __attribute__((aligned(16))) float Bm[FP * LDB];
    const int lane = threadIdx.x;
    const int r = lane & 15, q = lane >> 4;

    // Row pipeline (wmf_stream.h): factor rows are requested DEPTH groups ahead; the first loads of the
    // NEXT row are requested before this row's factorisation starts, so their latency hides behind it.
    constexpr int GS = 4, DEPTH = 3;
    using Stream = WmfRowStream<NFB, GS, DEPTH>;
    Stream st;
    int u = 0, d = 0;
    int64_t lo = 0;
    int64_t it = blockIdx.x;
    if (it < count) { u = rows[it]; lo = indptr[u]; d = (int)(indptr[u + 1] - lo); }
    auto prime = [&](int64_t lo_, int d_) {
        st.load_block(0, lo_, d_, indices, vals, lane, 0);
        st.load_block(1, lo_, d_, indices, vals, lane, 1);
        st.fetch_meta(0, q);
        st.template load_group<0>(0, V, ld, r, q);
        st.template load_group<1>(1, V, ld, r, q);
        st.template load_group<2>(2, V, ld, r, q);
    };
    if (it < count) prime(lo, d);

    for (; it < count; it += gridDim.x) {
        const int ngroups = (d + Stream::EPG - 1) / Stream::EPG;
        const int64_t itn = it + gridDim.x;
        int un = 0, dn = 0;
        int64_t lon = 0;
        if (itn < count) { un = rows[itn]; lon = indptr[un]; dn = (int)(indptr[un + 1] - lon); }

        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        float racc[NFB];
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) racc[fb] = 0.f;

        // consume ring slot S (group G), then refill it with group G + DEPTH
        auto step = [&](auto slot, int G) {
            constexpr int S = decltype(slot)::value;
            if (G >= ngroups) return;
            if (!(dbg & 2)) {
#pragma unroll
                for (int t = 0; t < GS; ++t) {
                    float fw[NFB];
                    st.template mask_tail<S>(t, ld, r);
#pragma unroll
                    for (int fb = 0; fb < NFB; ++fb) { fw[fb] = st.fr[S][t][fb] * st.w[S][t]; racc[fb] += st.fr[S][t][fb] * st.p[S][t]; }
                    int tt = 0;
#pragma unroll
                    for (int bi = 0; bi < NFB; ++bi)
#pragma unroll
                        for (int bj = bi; bj < NFB; ++bj, ++tt) acc[tt] = WMF_MFMA16(st.fr[S][t][bi], fw[bj], acc[tt]);
                }
            }
            const int next = G + DEPTH;
            if (next < ngroups) {
                if (next % Stream::GPB == 0) {                   // first group of block c: request block c + 1
                    const int c = next / Stream::GPB;
                    st.load_block(c + 1, lo, d, indices, vals, lane, (c + 1) & 1);
                }
                st.template load_group<S>(next, V, ld, r, q);
            }
        };
        for (int G0 = 0; G0 < ngroups; G0 += DEPTH) {
            step(std::integral_constant<int, 0>{}, G0);
            step(std::integral_constant<int, 1>{}, G0 + 1);
            step(std::integral_constant<int, 2>{}, G0 + 2);
        }
        // rhs: sum the four k-slot partials; every lane (r, *) then holds rhs[16 fb + r]
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) { racc[fb] += __shfl_xor(racc[fb], 16); racc[fb] += __shfl_xor(racc[fb], 32); }

        // ---- B: tiles (+ I) transposed into the lower triangle of Bm; b_i = rhs[i] on lane i
        {
            int tt = 0;
#pragma unroll
            for (int bi = 0; bi < NFB; ++bi) {
#pragma unroll
                for (int bj = bi; bj < NFB; ++bj, ++tt) {
                    float4 v = make_float4(acc[tt][0], acc[tt][1], acc[tt][2], acc[tt][3]);
                    if (bi == bj) {
                        if (r == 4 * q + 0) v.x += 1.f;
                        if (r == 4 * q + 1) v.y += 1.f;
                        if (r == 4 * q + 2) v.z += 1.f;
                        if (r == 4 * q + 3) v.w += 1.f;
                    }
                    *reinterpret_cast<float4*>(&Bm[(16 * bj + r) * LDB + 16 * bi + 4 * q]) = v;
                }
            }
        }
        float b = 0.f;
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) { const float v = __shfl(racc[fb], r); b = (q == fb) ? v : b; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (itn < count) prime(lon, dn);                         // in flight during the factorisation below

        // ---- C: left-looking blocked Cholesky; lane i owns row i while a panel is in registers
        bool ok = true;
        const int myr = min(lane, FP - 1);                       // lanes >= FP (f <= 48) shadow the last row, never store
        const float* myrow = Bm + myr * LDB;
#pragma unroll 1
        for (int p = 0; p < ((dbg & 1) ? 0 : NFB); ++p) {
            // C1. panel p (row tiles ib >= p) -= sum_{kb < p} L[ib][kb] L[p][kb]^T
            if (p > 0) {
#pragma unroll 1
                for (int ib = p; ib < NFB; ++ib) {
                    f32x4 c;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) c[reg] = Bm[(16 * ib + 4 * q + reg) * LDB + 16 * p + r];
#pragma unroll 1
                    for (int kb = 0; kb < p; ++kb) {
                        const float4 a4 = *reinterpret_cast<const float4*>(&Bm[(16 * ib + r) * LDB + 16 * kb + 4 * q]);
                        const float4 b4 = *reinterpret_cast<const float4*>(&Bm[(16 * p + r) * LDB + 16 * kb + 4 * q]);
                        c = WMF_MFMA16(-a4.x, b4.x, c); c = WMF_MFMA16(-a4.y, b4.y, c);
                        c = WMF_MFMA16(-a4.z, b4.z, c); c = WMF_MFMA16(-a4.w, b4.w, c);
                    }
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) Bm[(16 * ib + 4 * q + reg) * LDB + 16 * p + r] = c[reg];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            // C2. panel columns 16p .. 16p+15 of my row into registers
            float a[16];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const float4 v = *reinterpret_cast<const float4*>(myrow + 16 * p + 4 * c4);
                a[4 * c4] = v.x; a[4 * c4 + 1] = v.y; a[4 * c4 + 2] = v.z; a[4 * c4 + 3] = v.w;
            }
            // C3. factor the panel: column K = 16p + c; rows below K (all later rows included) get L[i][K];
            //     forward substitution fused: y_K = b_K / L[K][K], b_i -= L[i][K] y_K for i > K
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const int K = 16 * p + c;
                const float dk = rl64(a[c], K);
                if (!(dk > 1e-20f)) ok = false;
                const float inv = __builtin_amdgcn_rsqf(dk);
                a[c] *= inv;                                     // lane K: sqrt(dk); lanes > K: L[i][K]
                const float yk = rl64(b, K) * inv;
                b = (lane == K) ? yk : (lane > K ? b - a[c] * yk : b);
#pragma unroll
                for (int c2 = c + 1; c2 < 16; ++c2) a[c2] -= a[c] * rl64(a[c], 16 * p + c2);
            }
            // C4. panel back to LDS (rows above the panel carry junk into the never-read upper triangle)
            if (lane < FP) {
                float* dst = Bm + lane * LDB + 16 * p;
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4)
                    *reinterpret_cast<float4*>(dst + 4 * c4) = make_float4(a[4 * c4], a[4 * c4 + 1], a[4 * c4 + 2], a[4 * c4 + 3]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        // ---- D: backward substitution L^T g = y, lane j owns unknown j: for K descending
        //         g_K = y_K / L[K][K];  y_j -= L[K][j] g_K for j < K   (row K of L: consecutive lanes, no conflicts)
        float y = b;
        if (!(dbg & 1)) {
#pragma unroll 4
            for (int K = FP - 1; K >= 0; --K) {
                const float lkj = Bm[K * LDB + myr];             // lane K reads the diagonal
                const float gk = rl64(y, K) * __builtin_amdgcn_rcpf(rl64(lkj, K));
                y = (lane == K) ? gk : (lane < K ? y - lkj * gk : y);
            }
        }
        if (!ok) {                                               // not positive definite: the pivoted LU kernel redoes the row
            if (lane == 0) fb_rows[atomicAdd(fb_count, 1)] = u;
        } else {
            const int c = Stream::real_col(lane >> 4, lane & 15);        // undo the feature permutation of the row stream
            if (lane < FP && c < ld) g[(int64_t)u * ld + c] = (c < f) ? y : 0.f;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        u = un; lo = lon; d = dn;
    }
}

template <int NFB>
static void launch_direct64_nfb(const int32_t* rows, int64_t count, const float* V, const float* biasv,
                                const int64_t* indptr, const int32_t* indices, const float* vals, int f, int ld, float* g,
                                int32_t* fb_rows, int32_t* fb_count, int dbg, hipStream_t st) {
    int64_t grid = 256 * 12 * 2;                                 // ~12 waves per CU resident, two rounds queued
    if (grid > count) grid = count;
    hipLaunchKernelGGL((solve_direct64_kernel<NFB>), dim3((unsigned)grid), dim3(64), 0, st, rows, count, V, biasv, indptr,
                       indices, vals, f, ld, g, fb_rows, fb_count, dbg);
}

int wmf_launch_direct64(const int32_t* rows, int64_t count, const float* V, const float* biasv, const int64_t* indptr,
                        const int32_t* indices, const float* vals, int f, int ld, float* g, int32_t* fb_rows,
                        int32_t* fb_count, hipStream_t st) {
    if (count <= 0) return 0;
    const int dbg = wmf_debug_flags;
    switch ((f + 15) / 16) {
#define C_(N) case N: launch_direct64_nfb<N>(rows, count, V, biasv, indptr, indices, vals, f, ld, g, fb_rows, fb_count, dbg, st); break;
        C_(1) C_(2) C_(3) C_(4)
#undef C_
        default: return -1;
    }
    return 0;
}
