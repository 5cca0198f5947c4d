// Rows with more than 32 stored entries whose whitened system is close to the identity: a MATRIX-FREE polynomial iteration
// instead of forming and eliminating the f x f system (gfx950; round 4).
//
//   (I + E) g = b,   E = V_u^T D V_u,   b = V_u^T p          (RecModel/wmf_model.py:233-239 in whitened coordinates)
//
// The whitening bounds the whole fixed side by  sum_i v_i v_i^T <= I  (V = Y L^-T with L L^T = Y^T Y + lambda I), so a row that
// touches d of its m rows has  ||E||_2 <= tr E = sum_e w_e |v_e|^2 =: tau,  about  w d f / m  -- 0.013 for the item rows of
// BASELINE.json's configs[2] (d ~ 100 of m = 10^7 users), 0.13 / 0.26 for configs[1] / the one-GPU slice of configs[4].  Then
//     g = b - E b + E^2 b - ...                                         (tau <= tau_neumann; error after k terms <= tau^k)
// or the Chebyshev iteration on the spectrum bound [1 - tau_minus, 1 + tau_plus] (larger tau; tau_minus = the same sum over
// the negative weights a bias model can have) needs a handful of applications of E, and E d = V_u^T (D (V_u d)) is two passes
// over the GATHERED ROWS -- 2 d f multiply-adds -- where forming E costs d f^2 / 2 and eliminating it f^3 / 3: at f = 129,
// d = 100 four applications are 1 / 15 of the direct method's arithmetic, all of it plain f32 FMAs (no split-f16 operands: the
// result is accurate to f32 rounding of E d, better than the MFMA path).  The stopping test is on the true residual,
//     |r_k| phi <= eps |b| (1 - tau_minus),      phi = tau (Neumann) or delta / theta (Chebyshev),
// after which one more Richardson step x += r / theta costs nothing (its error is <= phi |A^-1 r|).  A row whose bound is too
// weak (kappa = (1 + tau_plus) / (1 - tau_minus) > kappa_max, tau_minus > 1/2) or that has not converged after kmax
// applications is BOUNCED: its id goes to a device-side list that the elimination kernels (wmf_directl / wmf_directw /
// wmf_rowsplit / wmf_wide) then solve as before.  Which path a row takes depends on its own data only, so results are
// reproducible bit for bit; the fraction of rows on each path is data dependent and is reported (wmf_plan_iter_stats).
//
// Work split.  One WORKGROUP of NW waves per row; the gathered rows live in REGISTERS from the first pass to the last:
// entry e = 4 NW s + 4 w + q of the row sits in slot s of wave w, lane group q = lane >> 4, whose 16 lanes r = lane & 15
// hold FPL consecutive features each (r FPL .. r FPL + FPL - 1; f <= 16 FPL, or 16 FPL + 1 with the border feature of the
// split layout, wmf_internal.h) -- a 16-lane group reads one whole row, 16-byte pieces, coalesced.  A vector of the
// iteration (x, r, d) is held in "feature layout": lane (q, r) has elements r FPL .., replicated over q and over the waves.
//   pass A   t_e = w_e (v_e . d): FPL / 2 packed FMAs per slot, then a sum over the 16 lanes of a group (four DPP adds that
//            serve the four entries of a slot at once);
//   pass B   z = sum_e t_e v_e: FPL / 2 packed FMAs per slot; the sum over the four lane groups is a reduce-scatter by lane
//            swaps (v_permlane32_swap / v_permlane16_swap: six instructions per four values), the sum over the waves goes
//            through LDS in a fixed order: one workgroup barrier per application of E.
// 4 waves x 3 workgroups per CU at f <= 144 (168 registers), 8 waves x 1 workgroup at f > 144; the next row's column ids are
// fetched a row ahead, so a row switch costs one memory latency, hidden by the other workgroups of the CU.
#include "wmf_common.h"
#include "wmf_internal.h"

typedef float it_f32x2 __attribute__((ext_vector_type(2)));


__device__ __forceinline__ int it_bits(float v) { return __builtin_bit_cast(int, v); }
__device__ __forceinline__ float it_flt(int v) { return __builtin_bit_cast(float, v); }

// lane (q, r) gets the sum over the four 16-lane groups of a_q -- a reduce-scatter: three swaps, three adds, fixed order
__device__ __forceinline__ float it_rs4(float a0, float a1, float a2, float a3) {
    const auto s02 = __builtin_amdgcn_permlane32_swap(it_bits(a0), it_bits(a2), false, false);   // {a0.lo a2.lo}, {a0.hi a2.hi}
    const float t02 = it_flt((int)s02[0]) + it_flt((int)s02[1]);
    const auto s13 = __builtin_amdgcn_permlane32_swap(it_bits(a1), it_bits(a3), false, false);
    const float t13 = it_flt((int)s13[0]) + it_flt((int)s13[1]);
    const auto u = __builtin_amdgcn_permlane16_swap(it_bits(t02), it_bits(t13), false, false);   // odd groups of t02 <-> even groups of t13
    return it_flt((int)u[0]) + it_flt((int)u[1]);
}

template <int NW, int FPL>
struct ItLds {
    static constexpr int FEAT = 16 * FPL;
    static constexpr int VEC = NW * FEAT;              // floats of one buffer of partial vectors
    static constexpr int EXCH = 2 * (VEC + NW * 4);    // two buffers (an exchange writes the one the previous did not)
};

// counters of a launch (per plan, wmf_plan_iter_stats): rows solved here, rows bounced, applications of E in total
enum { IT_STAT_DONE = 0, IT_STAT_BOUNCED = 1, IT_STAT_APPLICATIONS = 2, IT_STAT_CHEB = 3, IT_STAT_STAGE1 = 4 };   // ([4]: rows stage 1 handed to stage 2)


// ---- LDS access the compiler does not see (DMA variant): while an LDS-DMA (global_load_lds) may be outstanding hipcc puts
// s_waitcnt vmcnt(0) in front of every LDS access it knows of, which would stall each exchange of the iteration behind the
// prefetch of the next row.  Every LDS read / write of that variant is therefore inline asm with hand-placed waits.
typedef const __attribute__((address_space(1))) void* it_gptr;
typedef __attribute__((address_space(3))) void* it_lptr;
template <int OFF>
__device__ __forceinline__ void it_ds_write32(unsigned addr, float v) {
    asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ f32x4 it_ds_read128(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ float it_ds_read32(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// every outstanding LDS operation has returned; the values pass through the wait, so no use of them can move above it
// f(integral_constant<int, 0>) .. f(integral_constant<int, N - 1>): loops whose index must be a compile-time constant (immediates)
template <int N, class F>
__device__ __forceinline__ void it_for(F&& f) {
    [&]<int... Is>(std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }(std::make_integer_sequence<int, N>{});
}
#define IT_I(X) decltype(X)::value
__device__ __forceinline__ void it_lgkm_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void it_tie(f32x4& a) { asm volatile("" : "+v"(a)::"memory"); }
__device__ __forceinline__ void it_tie(float& a) { asm volatile("" : "+v"(a)::"memory"); }

// FULL: every lane's pieces lie inside a row (16 FPL floats per gathered row exactly: k = 64, 128, 256 with or without the
// split layout's border) -- no piece masks in the gather.
// DMA (four waves, FULL): the NEXT row's gathered rows, weights and border / bias values are fetched into an LDS region of the
// workgroup by LDS-DMA while this row is iterated on -- no registers, no wait until the row switch, where a wave copies its
// own entries from LDS to registers.  Without it a row switch exposes one memory latency per row (measured at configs[2]:
// 13.5 ms, of which 3 ms go when every gather hits the cache).
// LSB (split layout at 128 body floats, wmf_solve_rows_ex): the whitened side is in the ROLLED coordinates -- its border feature is
// the same number for every row (side[0]) -- and the fixed side's bias rides in the last mantissa bits of body features 8 j, 8 j + 1
// (wmf_row_transform mode 3): nothing is fetched from the pairs, whose 8 bytes cost a whole 128-byte line per stored entry.
template <int NW, int FPL, int NS, bool SPLIT, bool FULL, int OCC, bool DMA, bool LSB = false>
__global__ __launch_bounds__(64 * NW, OCC) void solve_iter_kernel(const int32_t* __restrict__ rows, int64_t count,
                                                                 const float* __restrict__ V, const float* __restrict__ side,
                                                                 const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                                 const float* __restrict__ vals, int f, int ld, float* __restrict__ g,
                                                                 int32_t* __restrict__ bounce_rows, int32_t* __restrict__ bounce_count,
                                                                 float tau_neumann, float kappa_max, int kmax, float eps2,
                                                                 unsigned long long* __restrict__ stats, const int4* __restrict__ info,
                                                                 const int32_t* __restrict__ count_dev, int bounce_stat) {
    if (count_dev) count = *count_dev;                  // (stage 2: the rows stage 1 handed on; the count is on the device)
    constexpr int H = FPL / 2, P4 = FPL / 4, EPS = 4 * NW;
    // The two-wave geometry (stage 1 at f = 128 / 129) runs the Neumann series only -- a fourth vector does not fit its
    // registers; a row that needs the Chebyshev recurrence is handed to the four-wave kernel, which has it (wmf_launch_iter).
    constexpr bool CHEB = NW != 2;
    using L = ItLds<NW, FPL>;
    static_assert(!DMA || (NW == 4 && FULL), "DMA variant: four waves, whole pieces");
    static_assert(!LSB || (SPLIT && FULL && FPL == 8), "bias bits: two per lane of eight features");
    // DMA: [exchange buffers | ring: wave, slot, piece -> 1 KB (lane l at 16 l) | meta: wave -> weights, border, bias (64 dwords each)]
    constexpr int RING_OFF = L::EXCH * 4, META_OFF = RING_OFF + NW * NS * P4 * 1024;
    extern __shared__ __attribute__((aligned(16))) char dyn_lds[];
    __shared__ __attribute__((aligned(16))) float lds_static[DMA ? 4 : L::EXCH];
    float* lds = DMA ? reinterpret_cast<float*>(dyn_lds) : lds_static;
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)dyn_lds);   // LDS byte address (DMA variant)
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int ldv = FULL ? 16 * FPL : (SPLIT ? f - 1 : ld);   // floats between gathered rows (split layout: the packed body)
    const float m0 = (r == 0) ? 1.f : 0.f;              // the border feature is counted by one lane of a group
    const float cb = LSB ? side[0] : 0.f;               // LSB: every row's border value
    // LSB: the 32 bits of an entry's bias, two from each of the 16 lanes that hold its row (features 8 r, 8 r + 1), or-ed together
    auto lsb_bias = [&](const it_f32x2& v0) {
        int b = ((it_bits(v0[0]) & 1) | ((it_bits(v0[1]) & 1) << 1)) << (2 * r);
        b |= __builtin_amdgcn_update_dpp(0, b, 0xB1, 0xF, 0xF, true);      // quad_perm [1,0,3,2]
        b |= __builtin_amdgcn_update_dpp(0, b, 0x4E, 0xF, 0xF, true);      // quad_perm [2,3,0,1]
        b |= __builtin_amdgcn_update_dpp(0, b, 0x141, 0xF, 0xF, true);     // row_half_mirror
        b |= __builtin_amdgcn_update_dpp(0, b, 0x140, 0xF, 0xF, true);     // row_mirror
        return it_flt(b);
    };
    // 16-byte pieces of this lane that lie inside a row (f not a multiple of 64: the last lanes have fewer, or none)
    bool pin[P4];
#pragma unroll
    for (int j = 0; j < P4; ++j) pin[j] = FULL || r * FPL + 4 * j < ldv;

    unsigned long long st_done = 0, st_bounced = 0, st_apps = 0, st_cheb = 0;
    int parity = 0;                                     // exchange buffer to write next

    // ---- the next row's column ids, a row ahead ----------------------------------------------------------------
    int64_t it = blockIdx.x;
    int u = 0, d = 0;
    int64_t lo = 0;
    int idx[NS];
    // row i of the list: {first entry (two words), row id, entries} -- ONE 16-byte scalar load from the plan's table (info) where
    // rows[i] -> indptr[u], indptr[u + 1] are two dependent ones; requested two rows ahead, so no row waits on it
    auto row_of = [&](int64_t i, int& uu, int64_t& l, int& dd) {
        if (info) {
            const int4 ri = info[i];
            l = (int64_t)(((unsigned long long)(unsigned)ri.y << 32) | (unsigned)ri.x);
            uu = ri.z;
            dd = ri.w;
        } else {
            uu = rows[i];
            l = indptr[uu];
            dd = (int)(indptr[uu + 1] - l);
        }
    };
    // The column ids of a row travel as NID coalesced registers per wave -- lane l of register c holds the id of entry 64 c + l
    // (every wave loads the whole row's ids: two to four 256-byte loads) -- and are fetched a row (DMA variant: two rows) ahead.
    // A lane takes the ids of ITS entries (slot s: entry 4 NW s + 4 w + q) out of them with one ds_bpermute per slot: no
    // per-slot id registers live across a row, no LDS parking (round-4 lab: sixteen id registers carried across the row were
    // what spilled first, and a scratch store of a freshly loaded value waits for its load).
    constexpr int NID = (EPS * NS + 63) / 64;
    auto fetch_idc = [&](int64_t l, int dd, int (&ic)[NID]) {
#pragma unroll
        for (int c = 0; c < NID; ++c) {
            const int e = 64 * c + lane;
            ic[c] = indices[l + (e < dd ? e : 0)];      // (entries past the row's end: entry 0 again, an L1 hit; never used with weight)
#ifdef IT_LAB_SAMEROWS                                  // lab: every row gathers the same few rows (cache hits): the kernel without its
            ic[c] &= 1023;                              // memory latency (results are those of a different matrix)
#endif
        }
    };
    auto slot_ids = [&](const int (&ic)[NID], int (&id)[NS]) {
        it_for<NS>([&](auto S) {
            constexpr int s = IT_I(S), c = (EPS * s) >> 6;      // (EPS divides 64: a slot's entries sit in one register)
            id[s] = __builtin_amdgcn_ds_bpermute((((EPS * s) & 63) + 4 * wv + q) * 4, ic[c]);
        });
    };
    auto meta_id = [&](const int (&ic)[NID]) {           // the id of this lane's entry in the meta layout (lane l < 4 NS: entry 4 NW (l >> 2) + 4 w + (l & 3))
        const int e = (lane < 4 * NS) ? EPS * (lane >> 2) + 4 * wv + (lane & 3) : 0;
        int v = __builtin_amdgcn_ds_bpermute((e & 63) * 4, ic[0]);
#pragma unroll
        for (int c = 1; c < NID; ++c) {
            const int vc = __builtin_amdgcn_ds_bpermute((e & 63) * 4, ic[c]);
            v = (e >> 6) == c ? vc : v;
        }
        return v;
    };
    // DMA variant: the requests of one row (l, dd) whose ids are ic: per slot in use P4 instructions of 64 x 16 bytes, then one
    // dword instruction for the weights and two for the {border, bias} pairs.  Destinations are wave uniform (M0); a lane's data
    // lands at 16 l / 4 l.  (Every id is formed before the first request: an id register produced between two requests would be
    // waited for with vmcnt(0), i.e. behind the requests already issued.)
    auto issue_dma = [&](int64_t l, int dd, const int (&ic)[NID]) {
        if constexpr (DMA) {
            const int nsn = (dd + EPS - 1) / EPS;
            int idl[NS];
            slot_ids(ic, idl);
            int idm = (SPLIT && !LSB) ? meta_id(ic) : 0;
#pragma unroll
            for (int s = 0; s < NS; ++s) asm volatile("" : "+v"(idl[s]));
            asm volatile("" : "+v"(idm));
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s < nsn) {
                    const float* row = V + (int64_t)idl[s] * ldv + r * FPL;
#pragma unroll
                    for (int j = 0; j < P4; ++j)
                        __builtin_amdgcn_global_load_lds((it_gptr)(row + 4 * j), (it_lptr)(dyn_lds + RING_OFF + ((wv * NS + s) * P4 + j) * 1024), 16, 0, 0);
                }
            }
            if (lane < 4 * NS) {
                const int e = EPS * (lane >> 2) + 4 * wv + (lane & 3);
                __builtin_amdgcn_global_load_lds((it_gptr)(vals + l + (e < dd ? e : 0)), (it_lptr)(dyn_lds + META_OFF + wv * 768), 4, 0, 0);
                if constexpr (SPLIT && !LSB) {
                    __builtin_amdgcn_global_load_lds((it_gptr)(side + 2 * (int64_t)idm), (it_lptr)(dyn_lds + META_OFF + wv * 768 + 256), 4, 0, 0);
                    __builtin_amdgcn_global_load_lds((it_gptr)(side + 2 * (int64_t)idm + 1), (it_lptr)(dyn_lds + META_OFF + wv * 768 + 512), 4, 0, 0);
                }
            }
        }
    };
    // Pipeline of the bookkeeping (G = gridDim.x rows apart).  Without DMA: row i's trip holds its own ids (idc, loaded during
    // the trip before), requests those of row i + G and the record of row i + 2 G.  With DMA everything moves one row further
    // ahead, so that the fetch of row i + G can be requested as soon as row i has been copied out of the ring, a whole row
    // before it is needed: the trip of row i holds the ids of row i + G, requests those of row i + 2 G and the record of row i + 3 G.
    const int64_t G = gridDim.x;
    int un = 0, dn = 0, unn = 0, dnn = 0;               // the records of the next two rows
    int64_t lon = 0, lonn = 0;
    int idc[NID];                                       // ids carried into the next trip (no DMA: of the trip's own row; DMA: of the row after)
#pragma unroll
    for (int c = 0; c < NID; ++c) idc[c] = 0;
    if (it + G < count) row_of(it + G, un, lon, dn);
    if (DMA && it + 2 * G < count) row_of(it + 2 * G, unn, lonn, dnn);
    if (it < count) {
        row_of(it, u, lo, d);
        fetch_idc(lo, d, idc);
        if constexpr (DMA) {
            issue_dma(lo, d, idc);
            if (it + G < count) fetch_idc(lon, dn, idc);
        }
    }

    for (; it < count; it += gridDim.x) {
        const int ns = (d + EPS - 1) / EPS;             // slots in use (wave uniform)
        const int ns4 = (ns + 3) & ~3;                  // ... rounded up to the groups of four the passes work in
        if constexpr (!DMA) slot_ids(idc, idx);
        // ---- gather: REQUESTS ONLY (nothing in this loop reads what it loads, so no wait separates the slots' requests):
        // body pieces, weight, border / bias pair of every entry of this wave
        it_f32x2 vb[NS][H];
        float vbd[NS], wt[NS], bsv[NS];                     // bsv: the fixed side's bias of the entry, folded into the weight in pass 0
        // (first the zero entries that fill the last group of four -- a register write behind the requests would wait for them)
#pragma unroll
        for (int s = 1; s < NS; ++s) {
            if (s >= ns && s < ns4) {
#pragma unroll
                for (int j = 0; j < H; ++j) vb[s][j] = it_f32x2{0.f, 0.f};
                wt[s] = 0.f;
                vbd[s] = 0.f; bsv[s] = 0.f;
            }
        }
        if constexpr (DMA) {
            // the row was fetched into LDS while the previous one was iterated on: every DMA of this wave has landed after the
            // wait; a lane copies its own 16-byte pieces and its group's weight / border / bias to registers
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned ring_rd = lds0 + RING_OFF + wv * NS * P4 * 1024 + lane * 16;
            const unsigned meta_rd = lds0 + META_OFF + wv * 768 + q * 4;
            it_for<NS>([&](auto S) {
                constexpr int s = IT_I(S);
                if (s < ns) {
                    it_for<P4>([&](auto J) {
                        constexpr int j = IT_I(J);
                        const f32x4 piece = it_ds_read128<(s * P4 + j) * 1024>(ring_rd);
                        vb[s][2 * j] = it_f32x2{piece[0], piece[1]};
                        vb[s][2 * j + 1] = it_f32x2{piece[2], piece[3]};
                    });
                    wt[s] = it_ds_read32<16 * s>(meta_rd);
                    if constexpr (SPLIT && !LSB) { vbd[s] = it_ds_read32<256 + 16 * s>(meta_rd); bsv[s] = it_ds_read32<512 + 16 * s>(meta_rd); }
                    if constexpr (LSB) { vbd[s] = cb; bsv[s] = 0.f; }
                }
            });
            it_lgkm_wait();
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#pragma unroll
                for (int j = 0; j < H; ++j) asm volatile("" : "+v"(vb[s][j])::"memory");
                it_tie(wt[s]);
                if constexpr (SPLIT && !LSB) { it_tie(vbd[s]); it_tie(bsv[s]); }
            }
        } else {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s < ns) {
                const int e = EPS * s + 4 * wv + q;
                const float* row = V + (int64_t)idx[s] * ldv + r * FPL;
#pragma unroll
                for (int j = 0; j < P4; ++j) {
                    const f32x4 piece = *reinterpret_cast<const f32x4*>((FULL || pin[j]) ? row + 4 * j : V);
                    vb[s][2 * j] = it_f32x2{piece[0], piece[1]};
                    vb[s][2 * j + 1] = it_f32x2{piece[2], piece[3]};
                }
                wt[s] = vals[lo + (e < d ? e : 0)];
                if constexpr (SPLIT && !LSB) { vbd[s] = side[2 * (int64_t)idx[s]]; bsv[s] = side[2 * (int64_t)idx[s] + 1]; }   // {last feature, bias}
                if constexpr (LSB) { vbd[s] = cb; bsv[s] = 0.f; }
            }
        }
        }
        const int64_t itn = it + G;
        int u3 = 0, d3 = 0;                                 // the record requested in this trip
        int64_t lo3 = 0;
        if constexpr (DMA) {
            // the ring is free (this wave's copies to registers have returned): the NEXT row's fetch starts here and has the whole
            // of this row's passes to land.  (The ids are read unconditionally first: their loads sat under a test like the
            // requests do, but the compiler does not correlate the two, and on the path "loaded, not requested" that it sees the
            // id registers would still be in flight at the join -- where the first instruction to reuse one got an
            // s_waitcnt vmcnt(0), i.e. a wait for the whole prefetch right behind its issue.)
#pragma unroll
            for (int c = 0; c < NID; ++c) asm volatile("" : "+v"(idc[c]));
            if (itn < count) issue_dma(lon, dn, idc);
            // ids of the row after the next, record of the one after that
            if (itn + G < count) fetch_idc(lonn, dnn, idc);
            if (itn + 2 * G < count) row_of(itn + 2 * G, u3, lo3, d3);
        } else {
            // the row after this one: its ids requested now (they stay in idc until the next trip)
            if (itn < count) fetch_idc(lon, dn, idc);
            if (itn + G < count) row_of(itn + G, u3, lo3, d3);
        }

        // ---- one exchange: z (feature partials of this wave's entries) and NSC scalars -> totals over the row --------------
        auto exchange = [&]<int NSC>(std::integral_constant<int, NSC>, it_f32x2 (&z)[H], float& zb, float& s1, float& s2) {
            if constexpr (DMA) {
                // the same exchange with every LDS access in inline asm (see it_ds_write32): byte addresses, immediates for the
                // piece / wave offsets, lgkmcnt waits by hand, a bare s_barrier (no fence: nothing but LDS is shared)
                constexpr int BUF = (L::VEC + NW * 4) * 4;
                const unsigned base = lds0 + parity * BUF;
                const unsigned wr = base + (wv * L::FEAT + r * FPL + q) * 4, scw = base + L::VEC * 4 + (wv * 4 + q) * 4;
                const unsigned rd = base + r * FPL * 4, scr = base + L::VEC * 4;
                it_for<P4>([&](auto J) {
                    constexpr int j = IT_I(J);
                    it_ds_write32<16 * j>(wr, it_rs4(z[2 * j][0], z[2 * j][1], z[2 * j + 1][0], z[2 * j + 1][1]));
                });
                if constexpr (NSC > 0) {
                    const float tsc = NSC > 1 ? it_rs4(zb, s1, s2, 0.f) : wmf_qsum(zb);
                    if (r == 0 && (NSC > 1 || q == 0)) it_ds_write32<0>(scw, tsc);
                }
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                f32x4 part[NW][P4], sacc[NW];
                float sac1[NW];
                it_for<NW>([&](auto W) {
                    constexpr int w2 = IT_I(W);
                    it_for<P4>([&](auto J) {
                        constexpr int j = IT_I(J);
                        part[w2][j] = it_ds_read128<(w2 * L::FEAT + 4 * j) * 4>(rd);
                    });
                    if constexpr (NSC > 1) sacc[w2] = it_ds_read128<16 * w2>(scr);
                    else if constexpr (NSC == 1) sac1[w2] = it_ds_read32<16 * w2>(scr);
                });
                it_lgkm_wait();
#pragma unroll
                for (int w2 = 0; w2 < NW; ++w2) {
#pragma unroll
                    for (int j = 0; j < P4; ++j) it_tie(part[w2][j]);
                    if constexpr (NSC > 1) it_tie(sacc[w2]);
                    else if constexpr (NSC == 1) it_tie(sac1[w2]);
                }
#pragma unroll
                for (int w2 = 1; w2 < NW; ++w2) {                // (a fixed order: wave 0, 1, ..)
#pragma unroll
                    for (int j = 0; j < P4; ++j) part[0][j] += part[w2][j];
                    if constexpr (NSC > 1) sacc[0] += sacc[w2];
                    else if constexpr (NSC == 1) sac1[0] += sac1[w2];
                }
#pragma unroll
                for (int j = 0; j < P4; ++j) { z[2 * j] = it_f32x2{part[0][j][0], part[0][j][1]}; z[2 * j + 1] = it_f32x2{part[0][j][2], part[0][j][3]}; }
                if constexpr (NSC > 1) { zb = sacc[0][0]; s1 = sacc[0][1]; s2 = sacc[0][2]; }
                else if constexpr (NSC == 1) zb = sac1[0];
                parity ^= 1;
                return;
            }
            float* vec = lds + parity * (L::VEC + NW * 4);
            float* sc = vec + L::VEC;
#pragma unroll
            for (int j = 0; j < P4; ++j) {
                const float tot = it_rs4(z[2 * j][0], z[2 * j][1], z[2 * j + 1][0], z[2 * j + 1][1]);   // feature r FPL + 4 j + q
                vec[wv * L::FEAT + r * FPL + 4 * j + q] = tot;
            }
            if constexpr (NSC > 0) {
                const float tsc = NSC > 1 ? it_rs4(zb, s1, s2, 0.f) : wmf_qsum(zb);
                if (r == 0 && (NSC > 1 || q == 0)) sc[wv * 4 + q] = tsc;
            }
            __syncthreads();
            // (where registers allow -- four waves at two workgroups per CU -- every partial vector is requested before the first is
            // added; the sums are in wave order either way)
            constexpr bool BATCH = NW == 4 && OCC == 2;
            f32x4 acc[P4], part[BATCH ? 3 : 1][P4];
#pragma unroll
            for (int j = 0; j < P4; ++j) acc[j] = *reinterpret_cast<const f32x4*>(vec + r * FPL + 4 * j);
            if constexpr (BATCH) {
#pragma unroll
                for (int w2 = 1; w2 < 4; ++w2)
#pragma unroll
                    for (int j = 0; j < P4; ++j) part[w2 - 1][j] = *reinterpret_cast<const f32x4*>(vec + w2 * L::FEAT + r * FPL + 4 * j);
            }
            if constexpr (NSC > 1) {
                f32x4 sacc = *reinterpret_cast<const f32x4*>(sc);
#pragma unroll
                for (int w2 = 1; w2 < NW; ++w2) sacc += *reinterpret_cast<const f32x4*>(sc + 4 * w2);
                zb = sacc[0]; s1 = sacc[1]; s2 = sacc[2];
            } else if constexpr (NSC == 1) {
                float a = sc[0];
#pragma unroll
                for (int w2 = 1; w2 < NW; ++w2) a += sc[4 * w2];
                zb = a;
            }
            if constexpr (BATCH) {
#pragma unroll
                for (int w2 = 1; w2 < 4; ++w2)                   // (a fixed order: wave 0, 1, ..)
#pragma unroll
                    for (int j = 0; j < P4; ++j) acc[j] += part[w2 - 1][j];
            } else {
#pragma unroll
                for (int w2 = 1; w2 < NW; ++w2)
#pragma unroll
                    for (int j = 0; j < P4; ++j) acc[j] += *reinterpret_cast<const f32x4*>(vec + w2 * L::FEAT + r * FPL + 4 * j);
            }
#pragma unroll
            for (int j = 0; j < P4; ++j) { z[2 * j] = it_f32x2{acc[j][0], acc[j][1]}; z[2 * j + 1] = it_f32x2{acc[j][2], acc[j][3]}; }
            parity ^= 1;
        };
        // sum of squares of a vector in feature layout (the same bits on every lane of every wave), as a wave-uniform value
        auto norm2 = [&](const it_f32x2 (&v)[H], float vbr) {
            it_f32x2 a = v[0] * v[0];
            if constexpr (H > 1) {
                it_f32x2 a2 = v[1] * v[1];
#pragma unroll
                for (int j = 2; j < H; j += 2) { a = v[j] * v[j] + a; a2 = v[j + 1] * v[j + 1] + a2; }
                a += a2;
            }
            float t = a[0] + a[1];
            if constexpr (SPLIT) t = __builtin_fmaf(m0 * vbr, vbr, t);
            return it_flt(__builtin_amdgcn_readfirstlane(it_bits(wmf_row16_sum(t))));
        };
        // z = E y = V_u^T (D (V_u y)) over this wave's entries (partials: the caller's exchange sums them): pass A, t_e = w_e (v_e . y),
        // and pass B, z += t_e v_e, four slots at a time -- four independent chains, one 16-lane sum for the four
        // (slots [S0, S0 + 4 NG) in ONE basic block: with two groups together the compiler has eight independent chains to
        // interleave -- each wave spends a third of its time waiting to issue a dependent instruction, and only two waves
        // share a SIMD)
        // TAIL: the row's last group -- the slots past the row's end (one to three of them; they hold zeros) are skipped, not computed
        auto apply_slots = [&]<int S0, int NG, bool TAIL>(std::integral_constant<int, S0>, std::integral_constant<int, NG>, std::bool_constant<TAIL>,
                                                          const it_f32x2 (&y)[H], float yb, it_f32x2 (&z)[H], float& zb) {
            constexpr int N = 4 * NG;
            it_f32x2 a[N];
            float tt[N];
            const float ybm = yb * m0;                           // the border product enters the 16-lane sum once
            if constexpr (TAIL) {
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    tt[i] = 0.f;
                    if (S0 + i < NS && S0 + i < ns) {
                        a[i] = vb[S0 + i < NS ? S0 + i : 0][0] * y[0];
#pragma unroll
                        for (int j = 1; j < H; ++j) a[i] = vb[S0 + i < NS ? S0 + i : 0][j] * y[j] + a[i];
                        tt[i] = a[i][0] + a[i][1];
                        if constexpr (SPLIT) tt[i] = __builtin_fmaf(vbd[S0 + i < NS ? S0 + i : 0], ybm, tt[i]);
                    }
                }
            } else {
#pragma unroll
            for (int i = 0; i < N; ++i) a[i] = (S0 + i < NS) ? vb[S0 + i < NS ? S0 + i : 0][0] * y[0] : it_f32x2{0.f, 0.f};
#pragma unroll
            for (int j = 1; j < H; ++j)
#pragma unroll
                for (int i = 0; i < N; ++i)
                    if (S0 + i < NS) a[i] = vb[S0 + i < NS ? S0 + i : 0][j] * y[j] + a[i];
#pragma unroll
            for (int i = 0; i < N; ++i) {
                tt[i] = a[i][0] + a[i][1];
                if constexpr (SPLIT) { if (S0 + i < NS) tt[i] = __builtin_fmaf(vbd[S0 + i < NS ? S0 + i : 0], ybm, tt[i]); }
            }
            }
#pragma unroll
            for (int g4 = 0; g4 < N; g4 += 4) wmf_row16_sum4(tt[g4], tt[g4 + 1], tt[g4 + 2], tt[g4 + 3]);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if (S0 + i < NS && (!TAIL || S0 + i < ns)) {
                    const int s = S0 + i;
                    const float t = tt[i] * wt[s];
#pragma unroll
                    for (int j = 0; j < H; ++j) z[j] = vb[s][j] * it_f32x2{t, t} + z[j];
                    if constexpr (SPLIT) zb = __builtin_fmaf(t, vbd[s], zb);
                }
            }
        };
        constexpr bool PAIRS = NW == 4 && OCC == 2;            // (where registers allow: not at eight waves, not at three workgroups per CU)
        auto apply = [&](const it_f32x2 (&y)[H], float yb, it_f32x2 (&z)[H], float& zb) {
            zb = 0.f;
#pragma unroll
            for (int j = 0; j < H; ++j) z[j] = it_f32x2{0.f, 0.f};
            it_for<(NS + 7) / 8>([&](auto P) {
                constexpr int s8 = 8 * IT_I(P);
                constexpr auto one = std::integral_constant<int, 1>{};
                constexpr auto two = std::integral_constant<int, 2>{};
                auto group = [&](auto S) {                          // one group of four, whole or the row's last
                    constexpr int s4 = IT_I(S);
                    if (s4 + 4 <= ns) apply_slots(S, one, std::false_type{}, y, yb, z, zb);
                    else apply_slots(S, one, std::true_type{}, y, yb, z, zb);
                };
                if (s8 < ns) {
                    if constexpr (PAIRS && s8 + 4 < NS) {
                        if (s8 + 8 <= ns) apply_slots(std::integral_constant<int, s8>{}, two, std::false_type{}, y, yb, z, zb);
                        else {
                            group(std::integral_constant<int, s8>{});
                            if (s8 + 4 < ns) group(std::integral_constant<int, s8 + 4>{});
                        }
                    } else {
                        group(std::integral_constant<int, s8>{});
                        if constexpr (s8 + 4 < NS) {
                            if (s8 + 4 < ns) group(std::integral_constant<int, s8 + 4>{});
                        }
                    }
                }
            });
        };

        // ---- pass 0 (the first reader of what the gather requested: slot by slot as the rows land): the entry's weight and
        // border value, b = V_u^T p, tau_plus / tau_minus = sum of |w| |v|^2 over the positive / negative weights
        it_f32x2 bv[H];
        float bb = 0.f, tp = 0.f, tn = 0.f;
#pragma unroll
        for (int j = 0; j < H; ++j) bv[j] = it_f32x2{0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s < ns) {
                const bool valid = EPS * s + 4 * wv + q < d;
                float w = wt[s];
                if constexpr (LSB) w -= lsb_bias(vb[s][0]);  // ... rebuilt from the gathered row's own bits
                else if constexpr (SPLIT) w -= bsv[s];      // the fixed side's bias comes with the row (RecModel/wmf_model.py:343)
                // (vbd: the same on the 16 lanes of a group -- its uses count it once, see ybm / m0)
                w = valid ? w : 0.f;
                wt[s] = w;
                const float pp = valid ? w + 1.f : 0.f;  // p = w + 1 (wmf_model.py:239); nothing for the lanes past the row's end
                if constexpr (!FULL) {
#pragma unroll
                    for (int j = 0; j < P4; ++j)
                        if (!pin[j]) { vb[s][2 * j] = it_f32x2{0.f, 0.f}; vb[s][2 * j + 1] = it_f32x2{0.f, 0.f}; }
                }
                it_f32x2 n2 = vb[s][0] * vb[s][0], n2b = vb[s][1] * vb[s][1];
#pragma unroll
                for (int j = 0; j < H; ++j) {
                    bv[j] = vb[s][j] * it_f32x2{pp, pp} + bv[j];
                    if (j >= 2) { if (j & 1) n2b = vb[s][j] * vb[s][j] + n2b; else n2 = vb[s][j] * vb[s][j] + n2; }
                }
                n2 += n2b;
                float nn = n2[0] + n2[1];
                if constexpr (SPLIT) {
                    bb = __builtin_fmaf(pp, vbd[s], bb);
                    nn = __builtin_fmaf(vbd[s] * m0, vbd[s], nn);   // (once per group: nn is summed over its 16 lanes)
                }
                tp = __builtin_fmaf(fmaxf(w, 0.f), nn, tp);
                tn = __builtin_fmaf(fmaxf(-w, 0.f), nn, tn);
            }
        }
        tp = wmf_row16_sum(tp);
        tn = wmf_row16_sum(tn);
        exchange(std::integral_constant<int, 3>{}, bv, bb, tp, tn);
        tp = it_flt(__builtin_amdgcn_readfirstlane(it_bits(tp)));
        tn = it_flt(__builtin_amdgcn_readfirstlane(it_bits(tn)));
        const float nb = norm2(bv, bb);

        // ---- the iteration's constants (wave uniform; the same on every wave of the workgroup) ----------------------------
        const float tau = tp + tn;
        const float alo = 1.f - tn, chi = 1.f + tp;             // the spectrum of I + E lies in [alo, chi]
        // (v_rcp_f32 / v_sqrt_f32 are accurate to one ulp: these constants steer the recurrence, they are not part of the answer)
        const float theta = 0.5f * (chi + alo), delta = 0.5f * (chi - alo), itheta = __builtin_amdgcn_rcpf(theta);
        const bool cheb_ok = tn <= 0.5f && chi <= kappa_max * alo;
        const float sk = __builtin_amdgcn_sqrtf(chi * __builtin_amdgcn_rcpf(fmaxf(alo, 0.25f)));
        const float sigma = (sk - 1.f) * __builtin_amdgcn_rcpf(sk + 1.f);   // the Chebyshev iteration's asymptotic rate on that interval
        const float stop = eps2 * nb * alo * alo;
        bool converged = false, cheb = !(tau <= tau_neumann);
        const bool go = cheb ? (CHEB && cheb_ok) : true;
        int napp = 0;

        it_f32x2 xv[H], rv[H];
        float xb = bb, rb = bb;
#pragma unroll
        for (int j = 0; j < H; ++j) { xv[j] = bv[j]; rv[j] = bv[j]; }
        if (go && !cheb) {
            // Neumann: x = sum_k (-E)^k b.  rv holds y = E^k b; the residual of the x BEFORE a term is added is that term.
            float sign = -1.f, nprev = nb;
            for (; napp < kmax;) {
                it_f32x2 z[H];
                float zb, z1 = 0.f, z2 = 0.f;
                apply(rv, rb, z, zb);
                exchange(std::integral_constant<int, SPLIT ? 1 : 0>{}, z, zb, z1, z2);
                ++napp;
                const float nr = norm2(z, zb);
                if (nr * tau * tau <= stop) {                   // adding this term leaves an error <= tau |A^-1 z|
#pragma unroll
                    for (int j = 0; j < H; ++j) xv[j] = z[j] * it_f32x2{sign, sign} + xv[j];
                    if constexpr (SPLIT) xb = __builtin_fmaf(sign, zb, xb);
                    converged = true;
                    break;
                }
                // contracting slowly (the row's operator has an eigenvalue near its bound tau): the Chebyshev recurrence on
                // [alo, chi] does better from here, started from the present x with this term as its residual
                if (cheb_ok && nr > fmaxf(sigma * sigma, 0.01f) * nprev) {
                    if constexpr (!CHEB) break;                 // (handed on: converged stays false)
#pragma unroll
                    for (int j = 0; j < H; ++j) rv[j] = z[j] * it_f32x2{sign, sign};
                    if constexpr (SPLIT) rb = sign * zb;
                    cheb = true;
                    break;
                }
#pragma unroll
                for (int j = 0; j < H; ++j) { xv[j] = z[j] * it_f32x2{sign, sign} + xv[j]; rv[j] = z[j]; }
                if constexpr (SPLIT) { xb = __builtin_fmaf(sign, zb, xb); rb = zb; }
                sign = -sign;
                nprev = nr;
            }
        } else if (go) {
#pragma unroll
            for (int j = 0; j < H; ++j) xv[j] = it_f32x2{0.f, 0.f};   // Chebyshev from x = 0, r = b
            xb = 0.f;
        }
        if (CHEB && go && cheb && !converged && cheb_ok) {
            // Chebyshev iteration for (I + E) x = b on [alo, chi] from (x, r): d_0 = r / theta, then
            //   x += d;  r -= (I + E) d;  rho' = 1 / (2 sigma_1 - rho);  d = rho' rho d + (2 rho' / delta) r        (sigma_1 = theta / delta)
            it_f32x2 dv[H];
            float db = rb * itheta, rho0 = delta * itheta;
            const float phi = delta * itheta;
            auto x_add = [&](const it_f32x2 (&dx)[H], float dxb, float sc) {      // x += sc dx
#pragma unroll
                for (int j = 0; j < H; ++j) xv[j] = dx[j] * it_f32x2{sc, sc} + xv[j];
                if constexpr (SPLIT) xb = __builtin_fmaf(sc, dxb, xb);
            };
#pragma unroll
            for (int j = 0; j < H; ++j) dv[j] = rv[j] * it_f32x2{itheta, itheta};
            for (; napp < kmax;) {
                x_add(dv, db, 1.f);
                it_f32x2 z[H];
                float zb, z1 = 0.f, z2 = 0.f;
                apply(dv, db, z, zb);
                exchange(std::integral_constant<int, SPLIT ? 1 : 0>{}, z, zb, z1, z2);
                ++napp;
#pragma unroll
                for (int j = 0; j < H; ++j) rv[j] = rv[j] - dv[j] - z[j];
                if constexpr (SPLIT) rb = rb - db - zb;
                const float nr = norm2(rv, rb);
                if (nr * phi * phi <= stop) {           // one Richardson step more is free: its error is <= phi |A^-1 r|
                    x_add(rv, rb, itheta);
                    converged = true;
                    break;
                }
                const float irho = __builtin_amdgcn_rcpf(2.f * theta - rho0 * delta);   // rho_1 / delta, rho_1 = 1 / (2 sigma_1 - rho_0)
                const float rho1 = delta * irho;
                const float alpha = 2.f * irho, beta = rho1 * rho0;
                rho0 = rho1;
#pragma unroll
                for (int j = 0; j < H; ++j) dv[j] = rv[j] * it_f32x2{alpha, alpha} + dv[j] * it_f32x2{beta, beta};
                if constexpr (SPLIT) db = alpha * rb + beta * db;
            }
        }
        if (converged) {
            if (wv == 0 && q == 0) {
                float* out = g + (int64_t)u * ld;
#pragma unroll
                for (int j = 0; j < P4; ++j) {
                    const int c = r * FPL + 4 * j;
                    if (FULL || c < ldv) *reinterpret_cast<f32x4*>(out + c) = f32x4{xv[2 * j][0], xv[2 * j][1], xv[2 * j + 1][0], xv[2 * j + 1][1]};
                }
                if (SPLIT && r == 0) *reinterpret_cast<f32x4*>(out + f - 1) = f32x4{xb, 0.f, 0.f, 0.f};
            }
            st_done++;
            st_apps += napp;
            st_cheb += cheb ? 1 : 0;
        } else {
            if (threadIdx.x == 0) bounce_rows[atomicAdd(bounce_count, 1)] = u;
            st_bounced++;
        }
        u = un; lo = lon; d = dn;
        if constexpr (DMA) {
            un = unn; lon = lonn; dn = dnn;
            unn = u3; lonn = lo3; dnn = d3;
        } else {
            un = u3; lon = lo3; dn = d3;
        }
    }
    if (stats && threadIdx.x == 0) {
        atomicAdd(stats + IT_STAT_DONE, st_done);
        atomicAdd(stats + bounce_stat, st_bounced);
        atomicAdd(stats + IT_STAT_APPLICATIONS, st_apps);
        atomicAdd(stats + IT_STAT_CHEB, st_cheb);
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------
// Geometry per width: waves per row NW, features per lane FPL (16 FPL floats of a gathered row), slots NS (a workgroup holds
// 4 NW NS entries), waves per SIMD OCC.  The split layout's extra registers (border value, pair) cost a slot or a workgroup.
//   ldv <=  64: 4 waves, FPL  4, 16 slots (256 entries), 3 workgroups per CU;  split: 12 slots (192 entries)
//   ldv <= 128: 4 waves, FPL  8,  9 slots (144 entries), 3 workgroups per CU;  split:  8 slots (128 entries)   (k = 128 +- biases)
//   ldv <= 192: 8 waves, FPL 12,  8 slots (256 entries), 1 workgroup per CU
//   ldv <= 256: 8 waves, FPL 16,  8 slots (256 entries), 1 workgroup per CU                                     (k = 256)
//   ldv <= 320: 8 waves, FPL 20,  6 slots (192 entries), 1 workgroup per CU                                     (f up to 272)
#ifndef IT_NS64
#define IT_NS64 16
#endif
#ifndef IT_NS64S
#define IT_NS64S 16
#endif
#ifndef IT_NS128
#define IT_NS128 9
#endif
#ifndef IT_NS128S
#define IT_NS128S 9
#endif
#ifndef IT_DMA
#define IT_DMA 1             // ldv = 64 / 128 exactly: the next row prefetched into LDS by LDS-DMA (8 slots: 128 entries, two workgroups per CU)
#endif

#ifndef IT_NW128
#define IT_NW128 2           // ldv = 128 in two stages: two waves per row (16 slots of 8 entries, four workgroups per CU, plain loads,
#endif                       // Neumann series only), then the four-wave DMA kernel over the rows that stage handed on.  4: the DMA kernel alone
#ifndef IT_OCC4
#define IT_OCC4 3            // waves per SIMD of the four-wave geometries: ldv <= 128 without the split layout ...
#endif
#ifndef IT_OCC64
#define IT_OCC64 2           // ... ldv <= 64 (16 slots) ...
#endif
#ifndef IT_OCC4S
#define IT_OCC4S 2            // (three workgroups per CU spill the pairs: a scratch store of a loaded value waits for the load)
#endif
static int it_ldv(int f, int ld, bool split) { return split ? f - 1 : ld; }
static float it_env(const char* name, float dflt) {
    const char* s = getenv(name);
    return s && *s ? (float)atof(s) : dflt;
}

// rows of up to this many entries are candidates (0: no kernel for this width)
int wmf_iter_dmax(int f, int ld, int split) {
    const int ldv = it_ldv(f, ld, split != 0);
    if (split && ldv > 128) return 0;
    // Narrow factors stay with the elimination kernels by default: at 4 features per lane the sums over the lanes of an entry
    // cost as much as its multiply-adds, and the f x f elimination is cheap -- measured on MI355X at k = 64 (BASELINE.json
    // configs[1], item rows of 200 entries, four applications of E): 1.17 ms against 1.05 ms for solve_directl.
    // WMF_ITER_MIN_LDV = 0 sends them here as well.
    static const int min_ldv = (int)it_env("WMF_ITER_MIN_LDV", 65.f);
    if (ldv < min_ldv) return 0;
    if (IT_NW128 == 2 && ldv == 128) return 8 * 16;
    if (IT_DMA && (ldv == 64 || ldv == 128) && !it_env("WMF_ITER_NO_DMA", 0.f)) return 16 * 8;
    if (ldv <= 64) return 16 * (split ? IT_NS64S : IT_NS64);
    if (ldv <= 128) return 16 * (split ? IT_NS128S : IT_NS128);
    if (ldv <= 256) return 32 * 8;
    if (ldv <= 320) return 32 * 6;
    return 0;
}

template <int NW, int FPL, int NS, bool SPLIT, bool FULL, int OCC, bool DMA = false, bool LSB = false>
static void it_launch(const int32_t* rows, int64_t count, const float* V, const float* side, const int64_t* indptr,
                      const int32_t* indices, const float* vals, int f, int ld, float* g, int32_t* bounce_rows,
                      int32_t* bounce_count, unsigned long long* stats, const int4* info, hipStream_t st,
                      const int32_t* count_dev = nullptr, int bounce_stat = IT_STAT_BOUNCED) {
    static const char* nm = wmf_kname("solve_iter_kernel<%d, %d, %d, %s, %s, %d, %s, %s>", NW, FPL, NS, SPLIT ? "true" : "false",
                                      FULL ? "true" : "false", OCC, DMA ? "true" : "false", LSB ? "true" : "false");      // (as rocprofv3 prints it)
    using L = ItLds<NW, FPL>;
    // (DMA variant: exchange buffers, the ring of the next row's gathered rows, its weights and border / bias values)
    constexpr size_t dyn = DMA ? (size_t)L::EXCH * 4 + (size_t)NW * NS * (FPL / 4) * 1024 + NW * 768 : 0;
    static bool attr_set = false;
    if (DMA && !attr_set) {
        (void)hipFuncSetAttribute((const void*)solve_iter_kernel<NW, FPL, NS, SPLIT, FULL, OCC, DMA, LSB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        attr_set = true;
    }
    // policy (environment overrides for experiments): start with the Neumann series while tau <= WMF_ITER_TAU (it converges
    // for tau < 1, at the rate of the row's LARGEST EIGENVALUE, usually far below tau) and move to the Chebyshev recurrence
    // when it contracts slower than that would; Chebyshev only while the bound on the condition number is <= WMF_ITER_KAPPA;
    // at most WMF_ITER_KMAX applications of E
    static const float tau_n = it_env("WMF_ITER_TAU", 0.8f), kap = it_env("WMF_ITER_KAPPA", 4.f);
    static const int kmax = (int)it_env("WMF_ITER_KMAX", 20.f);
    static const float eps = it_env("WMF_ITER_EPS", 1.2e-7f);     // relative accuracy of a solved row: 2^-23, one float32 ulp
    const int64_t resident = 256LL * (OCC * 4 / NW);              // workgroups the chip holds
    // four rounds queued (rows differ in length); over a device-side list -- usually empty -- one round: every workgroup of a
    // launch has to be scheduled before it can find that out, 0.10 ms for 2048 workgroups of 72 KB of LDS
    const int64_t cap = count_dev ? resident : resident * 4;
    WMF_LAUNCH(nm, (solve_iter_kernel<NW, FPL, NS, SPLIT, FULL, OCC, DMA, LSB>), dim3((unsigned)(count < cap ? count : cap)), dim3(64 * NW), dyn, st,
               rows, count, V, side, indptr, indices, vals, f, ld, g, bounce_rows, bounce_count, tau_n, kap, kmax, eps * eps, stats, info, count_dev, bounce_stat);
}

// rows[0 .. count): candidates (more than 32 and at most wmf_iter_dmax entries).  side: NULL, or the {last feature, bias}
// pairs of the split layout (V is then the packed body).  Rows that are not solved here are appended to bounce_rows.
int wmf_launch_iter(const int32_t* rows, int64_t count, const float* V, const float* side, const int64_t* indptr,
                    const int32_t* indices, const float* vals, int f, int ld, float* g, int32_t* bounce_rows,
                    int32_t* bounce_count, unsigned long long* stats, const void* info_v, hipStream_t st, int lsb) {
    const int4* info = static_cast<const int4*>(info_v);     // NULL, or {first entry lo, hi, row id, entries} of rows[i] (wmf_plan_create)
    if (count <= 0) return 0;
    const bool split = side != nullptr;
    const int ldv = it_ldv(f, ld, split);
#define IT_GO(NW, FPL, NS, SP, OCC)                                                                                                      \
    do {                                                                                                                                 \
        if (ldv == 16 * FPL) it_launch<NW, FPL, NS, SP, true, OCC>(rows, count, V, side, indptr, indices, vals, f, ld, g, bounce_rows, bounce_count, stats, info, st); \
        else it_launch<NW, FPL, NS, SP, false, OCC>(rows, count, V, side, indptr, indices, vals, f, ld, g, bounce_rows, bounce_count, stats, info, st);            \
    } while (0)
    if (split && ldv > 128) return -1;                            // (the split layout exists for f <= 144 only)
#if IT_NW128 == 2
    if (ldv == 128 && !it_env("WMF_ITER_ONE_STAGE", 0.f)) {
        // Stage 1: two waves per row -- half the per-row overhead of the four-wave form (exchanges, norms, vector updates are per
        // wave) and four rows in flight per CU: 11.1 against 12.2 ms on the item side of BASELINE.json's configs[2] -- at the price of
        // registers: no room for the Chebyshev recurrence's fourth vector.  What it does not solve (tau above the Neumann limit, a
        // series that contracts slowly, no convergence) goes to the SECOND HALF of bounce_rows, counted in bounce_count[1].
        // Stage 2: the four-wave LDS-DMA kernel over that list (count on the device; an empty list costs a few microseconds); what
        // IT hands back is the final list, bounce_rows[0 ..) / bounce_count[0], for the elimination kernels.
        int32_t* handed = bounce_rows + count;
        if (split && lsb) {            // (the rolled coordinates with the bias in the rows' own bits: nothing fetched from the pairs)
            it_launch<2, 8, 16, true, true, 2, false, true>(rows, count, V, side, indptr, indices, vals, f, ld, g, handed, bounce_count + 1, stats, info, st, nullptr, IT_STAT_STAGE1);
            it_launch<4, 8, 8, true, true, 2, true, true>(handed, count, V, side, indptr, indices, vals, f, ld, g, bounce_rows, bounce_count, stats, nullptr, st, bounce_count + 1);
            return 0;
        }
        if (split) it_launch<2, 8, 16, true, true, 2>(rows, count, V, side, indptr, indices, vals, f, ld, g, handed, bounce_count + 1, stats, info, st, nullptr, IT_STAT_STAGE1);
        else it_launch<2, 8, 16, false, true, 2>(rows, count, V, side, indptr, indices, vals, f, ld, g, handed, bounce_count + 1, stats, info, st, nullptr, IT_STAT_STAGE1);
        if (split) it_launch<4, 8, 8, true, true, 2, true>(handed, count, V, side, indptr, indices, vals, f, ld, g, bounce_rows, bounce_count, stats, nullptr, st, bounce_count + 1);
        else it_launch<4, 8, 8, false, true, 2, true>(handed, count, V, side, indptr, indices, vals, f, ld, g, bounce_rows, bounce_count, stats, nullptr, st, bounce_count + 1);
        return 0;
    }
#endif
#if IT_DMA
    if ((ldv == 64 || ldv == 128) && !it_env("WMF_ITER_NO_DMA", 0.f)) {
#define IT_GO_DMA(FPL, SP) it_launch<4, FPL, 8, SP, true, 2, true>(rows, count, V, side, indptr, indices, vals, f, ld, g, bounce_rows, bounce_count, stats, info, st)
        if (ldv == 64) { if (split) IT_GO_DMA(4, true); else IT_GO_DMA(4, false); }
        else { if (split) IT_GO_DMA(8, true); else IT_GO_DMA(8, false); }
#undef IT_GO_DMA
        return 0;
    }
#endif
    if (ldv <= 64) { if (split) IT_GO(4, 4, IT_NS64S, true, IT_OCC4S); else IT_GO(4, 4, IT_NS64, false, IT_OCC64); }
    else if (ldv <= 128) { if (split) IT_GO(4, 8, IT_NS128S, true, IT_OCC4S); else IT_GO(4, 8, IT_NS128, false, IT_OCC4); }
    else if (ldv <= 192) IT_GO(8, 12, 8, false, 2);
    else if (ldv <= 256) IT_GO(8, 16, 8, false, 2);
    else if (ldv <= 320) IT_GO(8, 20, 6, false, 2);
    else return -1;
#undef IT_GO
    return 0;
}
