// Heavy rows for f = 128 / 129 (k = 128 without / with biases): the one-wave-per-row kernel of wmf_directw.hip with
// the gathered factor rows streamed through an LDS ring by LDS-DMA instead of a ring of registers (gfx950).
//
//   g_u = (I + V_u^T D V_u)^-1 V_u^T p          (RecModel/wmf_model.py:233-239 in whitened coordinates)
//
// Why: at f = 129 the 36 accumulator tiles and the elimination leave room for a register ring of 24 rows per wave, one
// wave per SIMD; a quarter of the kernel's wave cycles were s_waitcnt on those loads (profiles/r01_pmc_sq_counters.txt).
// global_load_lds_dwordx4 needs no destination registers: a ring of four 16-entry slots (32 KB per wave, four waves
// per CU) keeps 48 to 64 rows in flight, requested three groups = 432 MFMAs ahead of their use.
//
// Data path of one row (lo, d), groups of 16 entries, blocks of 64:
//   metadata  indices / weights of block b go to LDS by two dword DMAs (lane l = entry 64 b + l), one block ahead;
//             the border feature V[idx][128] (f = 129) by a third one once the indices have landed.
//   rows      DMA instruction i of group g fetches 32 pieces of 16 bytes (features 0..127) of two rows into
//             slot[g % 4] + 1024 i: lanes 0..31 entry 2 i, lanes 32..63 entry 2 i + 1, so that no instruction's kilobyte
//             crosses a row; a lane half h = lane >> 5 needs the indices of entries h, 2 + h, .. 14 + h: four ds_read2_b32.
//   consume   k-step t takes entries 4 t + q (k-steps past the row's end are skipped); lane (r, q) reads pieces r and r + 16 of its row (the feature
//             permutation of wmf_stream.h: virtual block 4 j + e = feature 64 j + 4 r + e), its weight and its border value.
// Everything hipcc must not see is inline asm: while an LDS-DMA is in flight the compiler waits vmcnt(0) before any LDS
// read or use of an ordinary load it knows of, which would drain the ring.  The waits are counted by hand (vmcnt is in
// issue order): group g has landed when at most 8 x (groups issued after it) vector-memory operations are outstanding.
// The elimination keeps w_p in registers (no LDS vectors), so no compiler-visible LDS access is left in the kernel.
#include "wmf_common.h"
#include "wmf_internal.h"
#include "wmf_dw_elim.h"

#include <utility>

#ifndef DL_GJ_LDS
#define DL_GJ_LDS 0                 // multiplier column of the tile inverse: 1 = ds_bpermute, 0 = two VALU lane swaps
#endif
#define DL_R 4                      // ring slots = groups per 64-entry block (the group loop is unrolled by it)
// metadata behind the ring: idx[4][BLK], w[4][BLK], border[4][BLK], bias[4][BLK] (BLK = 4 GE entries per block, block b in
// buffer b & 3; block b + 1 is requested while groups of b - 1 are still being consumed: three live blocks) at byte offsets
// 0, DL_W, DL_BD, DL_BB (the fixed side's bias: the second float of the pairs in the split layout) = multiples of DL_MARR(GE)
// per width (NFB = 4: f = 64 / 65, NFB = 8: f = 128 / 129): bytes of the feature part of a row, of a 16-entry slot, of the ring
// GE = entries per group (ring slot): 16, or 8 for the two-waves-per-SIMD variant at NFB = 8 (half the ring, metadata blocks
// of 32 entries, 16-entry chunks with 16x16x16 MFMAs, w_p in LDS instead of registers -- 19 KB of LDS per wave)
#define DL_RB(NFB) (64 * (NFB))
#define DL_SLOTB(NFB, GE) ((GE) * DL_RB(NFB))
#define DL_METAB(NFB, GE) (DL_R * DL_SLOTB(NFB, GE))
#define DL_MARR(GE) (64 * (GE))     // bytes of one metadata array: four buffers of 4 GE entries
#define DL_LDSB(NFB, GE) (DL_METAB(NFB, GE) + 4 * DL_MARR(GE) + ((GE) == 8 ? 1024 : 0))

typedef const __attribute__((address_space(1))) void* dl_gptr;
typedef __attribute__((address_space(3))) void* dl_lptr;

template <int OFF>
__device__ __forceinline__ f32x4 dl_read128(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
template <int OFF>
__device__ __forceinline__ float dl_read32(unsigned addr) {
    float v;
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OFF0, int OFF1>
__device__ __forceinline__ f32x2 dl_read2(unsigned addr) {       // dwords at addr + 4 OFF0 and addr + 4 OFF1
    f32x2 v;
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(OFF0), "n"(OFF1) : "memory");
    return v;
}
template <int N>
__device__ __forceinline__ void dl_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// X6: the accumulation B += V_u^T D V_u in split f16 instead of f32 MFMAs.  Operands are scaled by DL_S sqrt(w) and split
// into two f16 parts x = hi + lo (22 significand bits; the f32 remainder is exact, its conversion rounds at 2^-22 of x);
// per tile and 32 entries three v_mfma_f32_16x16x32_f16 -- lo.hi, hi.lo, hi.hi, every product exact in f32, the dropped
// lo.lo below 2^-22 of the product -- replace eight v_mfma_f32_16x16x4_f32 at half the cycles each, and the split costs a
// lane 3 instructions per value (v_cvt_pk_f16_f32 for two, one v_fma_mix per value for the remainder).  The round before
// used three bf16 parts and six MFMAs (exact split, 9 instructions per value); measured on the CPU prototype
// (tests/scale/proto_split.py, tests/scale/proto_f16.py) the accumulated tile and the solved row are as accurate as with f32
// MFMAs either way: the tile's own f32 accumulation, not the 2^-22 of the operands, sets its error.
// Scale DL_S: operands can be scaled by a power of two before the split (tiles, right-hand side and border then all carry
// DL_S^2, which the elimination is told about: wmf_dw_elim.h).  It is 1: whitened factors are small (|v| ~ 1 / sqrt(rows of
// the fixed side)) and their low parts are f16 denormals, but what the row system I + V^T D V needs is ABSOLUTE accuracy
// against its unit diagonal, and a denormal is exact to 3e-8 -- measured, DL_S = 1 and DL_S = 64 give the same rows
// (tools/lab/diag_f16.py), and only at unit scale are the inverse pivot tiles of the elimination O(1), which their own
// split needs.  Robustness: an operand beyond the f16 range (|v sqrt(w)| >= 65520) becomes an infinity whose hi.hi and lo.hi
// products meet in the diagonal tile as inf - inf, and a negative weight has no square root: either way a NaN reaches the
// pivot test and the row is bounced to the pivoted kernel, like every system that is not positive definite.  A chunk = two
// ring slots = 32 entries; lane (r, q) takes entries 8 q .. 8 q + 7 of the chunk (the K index of its MFMA operands).
#ifndef DL_S
#define DL_S 1.f
#endif
#ifndef DL_F16T
#define DL_F16T 1            // the elimination's tile products as split-f16 MFMAs too (wmf_dw_elim.h); 0: f32 MFMAs
#endif
#ifndef DL_PEEL
#ifdef WMF_LAB
#define DL_PEEL 0            // (the lab build's "no accumulation" switch needs initialised accumulators)
#else
#define DL_PEEL 1
#endif
#endif
#ifndef DL_LOMODE
#define DL_LOMODE 0          // lab: 1 = no low parts, 2 = low parts negated
#endif
// MODE 0: rows (accumulate + eliminate).  MODE 1: SEGMENTS of rows with more than WMF_HEAVY_T entries (wmf_plan): accumulate a
// segment's entries and store its partial system in the layout of solve_directw_kernel's MODE 1, whose MODE 2 then adds a
// row's segments and eliminates (wmf_directw.hip).
template <int NFB, bool BORDER, bool X6, int GE = 16, int MODE = 0>
__global__ __launch_bounds__(64, (NFB <= 4 || GE == 8) ? 2 : 1) void solve_directl_kernel(const int32_t* __restrict__ rows, int64_t count, const float* __restrict__ V,
                                                              const float* __restrict__ side, const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                              const float* __restrict__ vals, int f, int ld, float* __restrict__ g,
                                                              int32_t* __restrict__ fb_rows, int32_t* __restrict__ fb_count, int dbg,
                                                              const int64_t* __restrict__ seg_lo, const int32_t* __restrict__ seg_d,
                                                              float* __restrict__ partial, const int32_t* __restrict__ count_dev) {
    static_assert(MODE == 0 || X6, "segments: split-f16 path only");
    // count_dev != NULL: the number of rows is on the device (the list of rows the iteration kernel bounced, wmf_iter.hip)
    if (count_dev) count = *count_dev;
    // side != NULL (BORDER only): the split layout of a bias model's fixed side (wmf_internal.h) -- V holds packed body rows
    // of f - 1 = 16 NFB floats (exactly the RB bytes a ring row takes), side the {last feature, bias} pairs
    const bool split = BORDER && side != nullptr;
    const int ldv = split ? f - 1 : ld;
    constexpr int NT = NFB * (NFB + 1) / 2;
    static_assert(GE == 16 || (GE == 8 && X6), "8-entry groups: split-f16 path only");
    constexpr int RB = DL_RB(NFB), DL_SLOT = DL_SLOTB(NFB, GE), DL_META = DL_METAB(NFB, GE);   // row bytes, slot bytes, metadata offset
    constexpr int BLK = 4 * GE, MBUF = 4 * BLK; // entries per metadata block, bytes of one buffer of a metadata array
    constexpr int DL_W = DL_MARR(GE), DL_BD = 2 * DL_MARR(GE), DL_BB = 3 * DL_MARR(GE);
    constexpr int KC = 2 * GE;                  // X6: entries per chunk (two slots) = K of its MFMAs
    constexpr int EL = KC / 4;                  // X6: entries per lane and chunk (lane (r, q): entries EL q .. EL q + EL - 1)
    constexpr int J = NFB / 4;                  // 16-byte pieces per lane and entry (pieces r, r + 16, ..)
    constexpr int PP = 4 * NFB;                 // pieces per row; one DMA instruction moves 64 / PP rows
    constexpr int EPI = 64 / PP, NI = GE / EPI; // entries per DMA instruction, DMA instructions per group
    static_assert(NI == 4 || NI == 8, "four or eight DMA instructions per group");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int r = lane & 15, q = lane >> 4;
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);   // LDS byte address of the region
    int baddr[4];
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) baddr[kq] = (r + 16 * kq) * 4;
    // LDS byte addresses of this lane's reads (inline asm below)
    const unsigned ring_rd = lds0 + q * RB + r * 16;                                // + slot * DL_SLOT + t * 4 RB + piece * 256
    const unsigned meta_rd = lds0 + DL_META + q * 4;                               // + (block & 3) * 256 + (group in block * 16 + 2 t) * 4
    // Lane-derived addresses of the request side are recomputed where they are used (from a lane id the compiler cannot see
    // through): kept live across the row loop they are what hipcc spills first, and a scratch reload is a VMEM operation --
    // returned in order, so the s_waitcnt in front of its use drains every LDS-DMA of the ring.
    // (v_mbcnt in a volatile asm, not threadIdx.x or the builtin: hipcc spills even v0, and a CSE'd builtin result as well)
    auto fresh_lane = [&]() {
        int l;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
        return l;
    };

    // The ring starts as zeros.  A chunk of the X6 path reads 32 ring positions even where the row has fewer entries left;
    // those positions are cancelled by weight 0, which only works on finite data: after this, whatever is stale in the ring
    // is an older row's real factor data, never an uninitialised bit pattern (a NaN there would send the row to the pivoted kernel).
    {
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < DL_LDSB(NFB, GE) / 1024; ++i)   // (the metadata buffers too: a stale index is then a valid row, 0)
            asm volatile("ds_write_b128 %0, %1" ::"v"(lds0 + i * 1024 + lane * 16), "v"(z) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    auto item = [&](int64_t i, int& u_, int64_t& lo_, int& d_) {     // work item i: a row, or (MODE 1) a segment
        if constexpr (MODE == 1) { u_ = 0; lo_ = seg_lo[i]; d_ = seg_d[i]; }
        else { u_ = rows[i]; lo_ = indptr[u_]; d_ = (int)(indptr[u_ + 1] - lo_); }
    };
    // metadata of block b of row (lo_, d_): lane l <- entry min(64 b + l, d_ - 1) (clamped entries get weight 0 at use)
    // (metadata of block b of the row whose first block uses buffer `base`: buffer (base + b) & 3 -- the buffers rotate
    // across rows, so that the next row's first block can be requested while this row's last one is still in use)
    auto issue_meta = [&](int b, int64_t lo_, int d_, int base) {
        const int ln = fresh_lane();
        const int64_t e = lo_ + min(BLK * b + ln, d_ - 1);
        const unsigned par = ((base + b) & 3) * MBUF;
        if (BLK == 64 || ln < BLK) {                           // (lane l writes dword l of the buffer: the buffer has BLK of them)
            __builtin_amdgcn_global_load_lds((dl_gptr)(indices + e), (dl_lptr)(smem + DL_META + par), 4, 0, 0);
            __builtin_amdgcn_global_load_lds((dl_gptr)(vals + e), (dl_lptr)(smem + DL_META + DL_W + par), 4, 0, 0);
        }
    };
    // border feature of the entries of block b (its indices have landed)
    auto issue_border = [&](int b, int base) {
        if constexpr (BORDER) {
            const unsigned par = ((base + b) & 3) * MBUF;
            const int ln = fresh_lane();
            float iv = dl_read32<0>(lds0 + DL_META + par + (ln & (BLK - 1)) * 4);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(iv)::"memory");     // the value passes through the wait: no use can move above it
            const int idx = __builtin_bit_cast(int, iv);
            if (BLK < 64 && ln >= BLK) {
                // (nothing to fetch for this lane: the buffer holds BLK entries)
            } else if (split) {
                // the pair {last feature, bias}: two dwords of one 8-byte word (the pairs of a million rows are 8 MB: L2 hits)
                __builtin_amdgcn_global_load_lds((dl_gptr)(side + 2 * (int64_t)idx), (dl_lptr)(smem + DL_META + DL_BD + par), 4, 0, 0);
                __builtin_amdgcn_global_load_lds((dl_gptr)(side + 2 * (int64_t)idx + 1), (dl_lptr)(smem + DL_META + DL_BB + par), 4, 0, 0);
            } else {
                __builtin_amdgcn_global_load_lds((dl_gptr)(V + (int64_t)idx * ld + 16 * NFB), (dl_lptr)(smem + DL_META + DL_BD + par), 4, 0, 0);
            }
        }
    };
    // the 16 rows of group gi (S = gi % 4 = its ring slot and its position in the block) -> 8 DMA instructions
    // (S = gi % 4, the group's ring slot and its position in its block: a run-time value, so that the chunk loop below is
    // ONE loop body -- with the slot as a template constant the two unrolled bodies got different register assignments for
    // the accumulators and hipcc moved ~170 of them between AGPRs and VGPRs on every trip)
    auto issue_rows = [&](int S, int gi, int base) {
        const unsigned par = ((base + (gi >> 2)) & 3) * MBUF + S * (GE * 4);
        const int ln = fresh_lane();
        const unsigned idx8_rd = lds0 + DL_META + (ln / PP) * 4;     // + (block & 3) * MBUF + group in block * 4 GE
        const int piece = (ln % PP) * 4;                             // first float of this lane's piece
        // this lane's rows: entries EPI i + h of the group, i = 0 .. NI - 1
        f32x2 i0 = dl_read2<0, EPI>(idx8_rd + par), i1 = dl_read2<2 * EPI, 3 * EPI>(idx8_rd + par);
        f32x2 i2 = f32x2{0.f, 0.f}, i3 = f32x2{0.f, 0.f};       // (not copies of i0: it is still in flight)
        if constexpr (NI == 8) {
            i2 = dl_read2<4 * EPI, 5 * EPI>(idx8_rd + par);
            i3 = dl_read2<6 * EPI, 7 * EPI>(idx8_rd + par);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3)::"memory");
        const float ids[8] = {i0[0], i0[1], i1[0], i1[1], i2[0], i2[1], i3[0], i3[1]};
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int idx = __builtin_bit_cast(int, ids[i]);
            __builtin_amdgcn_global_load_lds((dl_gptr)(V + (int64_t)idx * ldv + piece), (dl_lptr)(smem + S * DL_SLOT + i * 1024), 16, 0, 0);
        }
    };

    int u = 0, d = 0;
    int64_t lo = 0;
    int64_t it = blockIdx.x;
    // first requests of a row: metadata of block 0 (and 1), border of block 0, groups 0 .. 2.  Nothing else of this wave
    // is in flight when this runs, so the vmcnt(0) in it waits for the row's own first metadata only.
    auto prime = [&](int64_t lo_, int d_, int base, bool meta0_requested) {
        const int ng = (d_ + GE - 1) / GE;
        if (!meta0_requested) issue_meta(0, lo_, d_, base);
        dl_wait_vm<0>();
        issue_border(0, base);
        if (d_ > BLK) issue_meta(1, lo_, d_, base);
        issue_rows(0, 0, base);
        if (ng > 1) issue_rows(1, 1, base);
        if (ng > 2) issue_rows(2, 2, base);
        if constexpr (X6) { if (ng > 3) issue_rows(3, 3, base); }     // two whole chunks
    };
    int mb = 0;                                         // metadata buffer of the current row's block 0
    if (it < count) { item(it, u, lo, d); prime(lo, d, mb, false); }

    for (; it < count; it += gridDim.x) {
        const int ngroups = (d + GE - 1) / GE;
        const int64_t itn = it + gridDim.x;
        int un = 0, dn = 0;
        int64_t lon = 0;
        if (itn < count) item(itn, un, lon, dn);
        const int mbn = (mb + ((ngroups + 3) >> 2)) & 3;   // metadata buffer of the next row's block 0

        // (DL_PEEL: the two-waves form takes the row's first chunk in a body of its own whose MFMAs start from C = 0 -- 144
        // accumulator writes per row less; rows of the heavy bin have more than 32 entries, so that chunk always exists.
        // MODE 1 segments may be shorter than a chunk, but never empty.)
        constexpr bool PEEL = X6 && GE == 8 && DL_PEEL != 0;
        f32x4 acc[NT];
        if constexpr (!PEEL) {
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        float racc[NFB], bacc[BORDER ? NFB : 1];
        float cacc = 0.f, eacc = 0.f;
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) racc[fb] = 0.f;
#pragma unroll
        for (int fb = 0; fb < (BORDER ? NFB : 1); ++fb) bacc[fb] = 0.f;

        // ---- A: one unrolled trip = the four groups of a 64-entry block
        auto step = [&](auto slot, int G) {
            constexpr int S = decltype(slot)::value;
            if (G >= ngroups) return;
            // request group G + 3 (slot (S + 3) % 4); it is the first group of a block when S == 1
            const int gn = G + 3;
            if (gn < ngroups) {
                constexpr int SN = (S + 3) & 3;
                if constexpr (SN == 0) {
                    // metadata of block gn / 4 was requested 32 operations ago (its own group requests and three more)
                    dl_wait_vm<4 * NI>();
                    issue_border(gn >> 2, mb);
                    if (BLK * ((gn >> 2) + 1) < d) issue_meta((gn >> 2) + 1, lo, d, mb);
                }
                issue_rows(SN, gn, mb);
            }
            // the row's last group: the NEXT row's first metadata block is requested now (its buffer, the one behind this
            // row's last block, is free) and has a group of MFMAs to arrive before prime() waits for it
            const bool last = G == ngroups - 1;
            if (last && itn < count) issue_meta(0, lon, dn, mbn);
            // group G has landed when only the requests made after it are outstanding
            const int younger = min(3, ngroups - 1 - G);
            if (younger >= 3) dl_wait_vm<3 * NI>();
            else if (younger == 2) dl_wait_vm<2 * NI>();
            else if (younger == 1) dl_wait_vm<NI>();
            else if (itn < count) dl_wait_vm<2>();       // only the next row's two metadata requests are younger
            else dl_wait_vm<0>();
            if (WMF_ABL(dbg, 2)) return;
            const unsigned par = ((mb + (G >> 2)) & 3) * 256;
            const int nk = min(4, (d - 16 * G + 3) >> 2);        // k-steps of this group that hold entries of the row
            // LDS operands of k-step t + 1 are requested before the MFMAs of k-step t (only the first k-step of a group
            // waits for its own reads with nothing to do)
            f32x4 xa[2], xb[2];
            float wv[2], bf[2] = {0.f, 0.f}, bb[2] = {0.f, 0.f};
            auto request = [&](auto tc) {
                constexpr int t = decltype(tc)::value;
                xa[t & 1] = dl_read128<S * DL_SLOT + t * 4 * RB>(ring_rd);
                if constexpr (J == 2) xb[t & 1] = dl_read128<S * DL_SLOT + t * 4 * RB + 256>(ring_rd);
                else xb[t & 1] = f32x4{0.f, 0.f, 0.f, 0.f};
                wv[t & 1] = dl_read32<DL_W + (S * 16 + 4 * t) * 4>(meta_rd + par);
                if constexpr (BORDER) {
                    bf[t & 1] = dl_read32<DL_BD + (S * 16 + 4 * t) * 4>(meta_rd + par);
                    bb[t & 1] = dl_read32<DL_BB + (S * 16 + 4 * t) * 4>(meta_rd + par);      // (zeros, or stale finite data, without biases)
                }
            };
            auto kstep = [&](auto tc) {
                constexpr int t = decltype(tc)::value, B = t & 1;
                if (t > 0 && t >= nk) return;                    // wave-uniform: past the row's end (nk >= 1: k-step 0, whose reads are
                                                                 // requested unconditionally below, always runs and waits for them)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xa[B]), "+v"(xb[B]), "+v"(wv[B]), "+v"(bf[B]), "+v"(bb[B])::"memory");
                const f32x4 ya = xa[B], yb = xb[B];
                const float wraw = split ? wv[B] - bb[B] : wv[B], bfv = bf[B];
                if constexpr (t < 3) { if (t + 1 < nk) request(std::integral_constant<int, t + 1>{}); }
                const bool real = 16 * G + 4 * t + q < d;
                const float w = real ? wraw : 0.f, p = real ? wraw + 1.f : 0.f;
                const float x8[8] = {ya[0], ya[1], ya[2], ya[3], yb[0], yb[1], yb[2], yb[3]};
                float x[NFB], fw[NFB];
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) x[fb] = x8[fb];
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) { fw[fb] = x[fb] * w; racc[fb] += x[fb] * p; }
                if constexpr (BORDER) {
                    const float bw = bfv * w;
#pragma unroll
                    for (int fb = 0; fb < NFB; ++fb) bacc[fb] += x[fb] * bw;
                    cacc += bfv * bw;
                    eacc += bfv * p;
                }
                int tt = 0;
#pragma unroll
                for (int bi = 0; bi < NFB; ++bi) {
#pragma unroll
                    for (int bj = bi; bj < NFB; ++bj, ++tt) acc[tt] = WMF_MFMA16(x[bi], fw[bj], acc[tt]);
                }
            };
            request(std::integral_constant<int, 0>{});
            [&]<int... Ts>(std::integer_sequence<int, Ts...>) {
                (kstep(std::integral_constant<int, Ts>{}), ...);
            }(std::make_integer_sequence<int, 4>{});
        };
        // ---- A (X6): one chunk = groups G, G + 1 = ring slots S, S + 1 (S = 0 or 2)
        auto chunk = [&](int S, int G, auto first_c) {           // S = G & 2: ring slots S, S + 1
            constexpr bool FIRST = decltype(first_c)::value;
            if (G >= ngroups) return;
            const bool lastc = G + 2 >= ngroups;                 // the row's last chunk
            if (lastc && itn < count) issue_meta(0, lon, dn, mbn);
            // groups G, G + 1 have landed when only what was requested after them is outstanding: groups G + 2, G + 3
            const int younger = max(0, min(2, ngroups - G - 2));
            if (younger == 2) dl_wait_vm<2 * NI>();
            else if (younger == 1) dl_wait_vm<NI>();
            else if (itn < count) dl_wait_vm<2>();
            else dl_wait_vm<0>();
            // all LDS operands of the chunk into registers: 8 entries x 2 pieces, their weights and border values
            const unsigned par = ((mb + (G >> 2)) & 3) * MBUF + S * (GE * 4);
            const int lnc = fresh_lane();
            const unsigned ring_s = lds0 + (lnc >> 4) * EL * RB + (lnc & 15) * 16 + S * DL_SLOT;   // entries EL q + j of the chunk: + j * RB + piece * 256
            const unsigned meta6_rd = lds0 + DL_META + (lnc >> 4) * EL * 4;                       // their EL weights / border values
            f32x4 xr[EL][2], wq[2], bq[2], bbq[2];
            auto read_entry = [&](auto jc) {
                constexpr int jj = decltype(jc)::value;
                xr[jj][0] = dl_read128<jj * RB>(ring_s);
                if constexpr (J == 2) xr[jj][1] = dl_read128<jj * RB + 256>(ring_s);
                else xr[jj][1] = f32x4{0.f, 0.f, 0.f, 0.f};
            };
            [&]<int... Js>(std::integer_sequence<int, Js...>) {
                (read_entry(std::integral_constant<int, Js>{}), ...);
            }(std::make_integer_sequence<int, EL>{});
            wq[0] = dl_read128<DL_W>(meta6_rd + par);
            wq[1] = bq[0] = bq[1] = bbq[0] = bbq[1] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (EL == 8) wq[1] = dl_read128<DL_W + 16>(meta6_rd + par);
            if constexpr (BORDER) {
                bq[0] = dl_read128<DL_BD>(meta6_rd + par);
                bbq[0] = dl_read128<DL_BB>(meta6_rd + par);
                if constexpr (EL == 8) {
                    bq[1] = dl_read128<DL_BD + 16>(meta6_rd + par);
                    bbq[1] = dl_read128<DL_BB + 16>(meta6_rd + par);
                }
            }
            if constexpr (EL == 8) {
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(xr[0][0]), "+v"(xr[0][1]), "+v"(xr[1][0]), "+v"(xr[1][1]), "+v"(xr[2][0]), "+v"(xr[2][1]), "+v"(xr[3][0]),
                               "+v"(xr[3][1]), "+v"(xr[EL - 4][0]), "+v"(xr[EL - 4][1]), "+v"(xr[EL - 3][0]), "+v"(xr[EL - 3][1]), "+v"(xr[EL - 2][0]), "+v"(xr[EL - 2][1]),
                               "+v"(xr[EL - 1][0]), "+v"(xr[EL - 1][1]), "+v"(wq[0]), "+v"(wq[1]), "+v"(bq[0]), "+v"(bq[1]), "+v"(bbq[0]), "+v"(bbq[1])::"memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(xr[0][0]), "+v"(xr[0][1]), "+v"(xr[1][0]), "+v"(xr[1][1]), "+v"(xr[2][0]), "+v"(xr[2][1]), "+v"(xr[3][0]),
                               "+v"(xr[3][1]), "+v"(wq[0]), "+v"(bq[0]), "+v"(bbq[0])::"memory");
            }
            // the two slots are free again: request groups G + 4, G + 5 into them (the first of them opens a block when S == 0)
            if (G + 4 < ngroups) {
                if (S == 0) {
                    issue_border((G >> 2) + 1, mb);              // its indices are older than the groups just waited for
                } else {
                    if (BLK * ((G >> 2) + 2) < d) issue_meta((G >> 2) + 2, lo, d, mb);   // two blocks on: needed in the next trip
                }
                issue_rows(S, G + 4, mb);
                if (G + 5 < ngroups) issue_rows(S + 1, G + 5, mb);
            }
            if (WMF_ABL(dbg, 2)) return;
            // right-hand side and border on the VALU from the raw values; MFMA operands scaled by sqrt(w) and split.
            // Block column by block column: the split of column bj + 1 (VALU) has no dependence on the MFMAs of column bj,
            // and a bf16 MFMA leaves half of its cycles to the VALU.
            float wj[EL], pj[EL], swj[EL], bwj[EL];
#pragma unroll
            for (int j = 0; j < EL; ++j) {
                const bool real = GE * G + EL * (lnc >> 4) + j < d;
                const float wraw = split ? wq[j >> 2][j & 3] - bbq[j >> 2][j & 3] : wq[j >> 2][j & 3];
                wj[j] = real ? wraw : 0.f;
                pj[j] = real ? wraw + 1.f : 0.f;
                swj[j] = DL_S * __builtin_amdgcn_sqrtf(wj[j]);
                const float bfv = bq[j >> 2][j & 3];
                bwj[j] = bfv * wj[j];
                if constexpr (BORDER) { cacc += bfv * bwj[j]; eacc += bfv * pj[j]; }
            }
            // the two f16 parts of a block column: EL halves each (f16x8 for 32-entry chunks, f16x4 for 16-entry chunks)
            typedef _Float16 f16xe __attribute__((ext_vector_type(EL)));
            f16xe hi[NFB], lo3[NFB];
            auto split = [&](int bj) {                           // right-hand side, border and the two f16 parts of block bj
                float x[EL];
#pragma unroll
                for (int j = 0; j < EL; ++j) {
                    x[j] = xr[j][bj >> 2][bj & 3];
                    racc[bj] += x[j] * pj[j];
                    if constexpr (BORDER) bacc[bj] += x[j] * bwj[j];
                }
#if DL_LOMODE == 0
                // hi = RN_f16(RN_f32(x sqrt(w))), lo against the exact product: wmf_split4_scaled (wmf_common.h)
                const wmf_u32x4 s0 = wmf_split4_scaled(x[0], x[1], x[2], x[3], swj[0], swj[1], swj[2], swj[3]);
                if constexpr (EL == 8) {
                    const wmf_u32x4 s1 = wmf_split4_scaled(x[EL - 4], x[EL - 3], x[EL - 2], x[EL - 1], swj[EL - 4], swj[EL - 3], swj[EL - 2], swj[EL - 1]);
                    hi[bj] = __builtin_bit_cast(f16xe, wmf_u32x4{s0[0], s0[1], s1[0], s1[1]});
                    lo3[bj] = __builtin_bit_cast(f16xe, wmf_u32x4{s0[2], s0[3], s1[2], s1[3]});
                } else {
                    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
                    hi[bj] = __builtin_bit_cast(f16xe, u32x2_{s0[0], s0[1]});
                    lo3[bj] = __builtin_bit_cast(f16xe, u32x2_{s0[2], s0[3]});
                }
#else
#pragma unroll
                for (int j = 0; j < EL; ++j) {
#pragma clang fp contract(off)
                    const float sx = x[j] * swj[j];
                    const _Float16 h = (_Float16)sx;
                    hi[bj][j] = h;
                    lo3[bj][j] = DL_LOMODE == 1 ? (_Float16)0.f : (_Float16)((float)h - sx);
                }
#endif
            };
            // (A hand-pipelined version -- MFMAs of chunk c between the split units of chunk c + 1, ring reads two entry pairs
            // ahead behind counted lgkmcnt waits: tools/lab/wmf_directl_pipe_attempt.hip.txt -- computed the same rows and was
            // slower, 29.0 against 26.9 ms at the time.)
            // Block column by block column, the split of column bj + 1 in front of the MFMAs of column bj.  hipcc puts ~340 of the
            // chunk's ~440 VALU instructions in front of the first MFMA and 80 of the 108 MFMAs behind the last one; pinning
            // each tile's three MFMAs together with a slice of the next column's split (sched_barrier) interleaves them as
            // intended and changes NOTHING (22.05 vs 22.11 ms at cfg3), and sched_group_barrier patterns make hipcc separate
            // them completely: on this kernel VALU time, MFMA pipe time and waits simply add up (rocprofv3 --pmc, cfg3 item
            // side: VALU active 63 % of the SIMD cycles, SQ_VALU_MFMA_BUSY_CYCLES 19 % = 16 cycles per f16 MFMA, SQ_WAIT_ANY
            // 19 %), so what shortens the phase is fewer instructions, not their order.
            auto mfma3 = [&](const f16xe& al, const f16xe& ah, const f16xe& bl, const f16xe& bh, f32x4 c) {
                if constexpr (EL == 8) {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
                } else {
                    c = __builtin_amdgcn_mfma_f32_16x16x16f16(al, bh, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bl, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x16f16(ah, bh, c, 0, 0, 0);
                }
                return c;
            };
            split(0);
#pragma unroll
            for (int bj = 0; bj < NFB; ++bj) {
                if (bj + 1 < NFB) split(bj + 1);
#pragma unroll
                for (int bi = 0; bi <= bj; ++bi) {
                    const int tt = tile_w<NFB>(bi, bj);
                    acc[tt] = mfma3(lo3[bi], hi[bi], lo3[bj], hi[bj], FIRST ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[tt]);
                }
            }
        };
        if constexpr (PEEL) chunk(0, 0, std::true_type{});
        if constexpr (X6) {
#pragma unroll 1
            for (int G = PEEL ? 2 : 0; G < ngroups; G += 2) chunk(G & 2, G, std::false_type{});
        }
        for (int G0 = 0; G0 < (X6 ? 0 : ngroups); G0 += DL_R) {
            if constexpr (X6) {
            } else {
                [&]<int... Ss>(std::integer_sequence<int, Ss...>) {
                    (step(std::integral_constant<int, Ss>{}, G0 + Ss), ...);
                }(std::make_integer_sequence<int, DL_R>{});
            }
        }
        if (itn < count) prime(lon, dn, mbn, true);      // the next row's first entries fly during the elimination
        if constexpr (MODE == 1) {                       // this segment's partial system (solve_directw_kernel MODE 1's layout)
            const int ln = fresh_lane(), rr = ln & 15, qq = ln >> 4;
            float* out = partial + it * (int64_t)WMF_DW_PARTIAL(NFB, BORDER);
#pragma unroll
            for (int bi = 0; bi < NFB; ++bi) {
#pragma unroll
                for (int bj = bi; bj < NFB; ++bj) {
                    const int t = tile_w<NFB>(bi, bj);
                    float* to = out + tile_off<NFB>(bi, bj);
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        if (bi != bj) to[reg * 64 + ln] = acc[t][reg];
                        else if (4 * qq + reg <= rr) to[rr * (rr + 1) / 2 + 4 * qq + reg] = acc[t][reg];   // element (4q + reg, r)
                    }
                }
            }
            float* vo = out + WMF_DW_TILES(NFB);
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) vo[fb * 64 + ln] = racc[fb];
            if constexpr (BORDER) {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) vo[(NFB + fb) * 64 + ln] = bacc[fb];
                vo[2 * NFB * 64 + ln] = cacc;
                vo[(2 * NFB + 1) * 64 + ln] = eacc;
            }
            u = un; lo = lon; d = dn; mb = mbn;
            continue;
        }

        // ---- C, D: block elimination and backward pass (wmf_dw_elim.h), w_p in registers
        bool ok = true;
        float gb[NFB];
        float tb = 0.f;
        if constexpr (X6) {                              // tiles carry DL_S^2 (scaled operands); the VALU sums do not yet
            constexpr float S2 = DL_S * DL_S;
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) racc[fb] *= S2;
            if constexpr (BORDER) {
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) bacc[fb] *= S2;
                cacc *= S2; eacc *= S2;
            }
        }
        // (GE == 8: w_p / w^b_p of the pivots in LDS behind the metadata -- 64 registers the two-waves variant does not have)
        float* Wv = reinterpret_cast<float*>(smem + DL_META + 4 * DL_MARR(GE));
        dw_eliminate<NFB, BORDER, (DL_GJ_LDS != 0), (GE != 8), (X6 && DL_F16T != 0), (GE == 8)>(acc, racc, bacc, cacc, eacc, GE == 8 ? Wv : nullptr,
                                                          GE == 8 ? Wv + 16 * NFB : nullptr, r, q, baddr, dbg, gb, tb, ok,
                                                          X6 ? DL_S * DL_S : 1.f);
        if (!ok) {
            if (lane == 0) fb_rows[atomicAdd(fb_count, 1)] = u;
        } else if (q == 0) {
#pragma unroll
            for (int p = 0; p < NFB; ++p) {
                const int c = 64 * (p >> 2) + 4 * r + (p & 3);    // undo the feature permutation
                g[(int64_t)u * ld + c] = gb[p];
            }
            if constexpr (BORDER) {
                const int c = 16 * NFB + r;                       // the border column and the padding behind it
                if (c < ld) g[(int64_t)u * ld + c] = (r == 0) ? tb : 0.f;
            }
        }
        u = un; lo = lon; d = dn; mb = mbn;
    }
}

int wmf_directl_supported(int f, int ld) {
    return (f == 128 && ld == 128) || (f == 129 && ld == 132) || (f == 64 && ld == 64) || (f == 65 && ld == 68);
}

// the launch over the rows the iteration kernel bounced is a separate line of the per-kernel timing table
static const char* dl_name(const char* base, bool bounced) { return bounced ? wmf_kname("%s [bounced]", base) : base; }

int wmf_launch_directl(const int32_t* rows, int64_t count, const float* V, const float* side, const int64_t* indptr,
                       const int32_t* indices, const float* vals, int f, int ld, float* g, int32_t* fb_rows, int32_t* fb_count,
                       hipStream_t st, const int32_t* count_dev) {
    // (count_dev: NULL, or the device-side number of rows -- count is then the capacity of the list and sizes the grid)
    if (count <= 0) return 0;
    if (!wmf_directl_supported(f, ld)) return -1;
    const int nfb = f / 16;                                      // 4 or 8 (the bias column is a border)
    const int64_t cap = 256 * 4 * (nfb <= 4 ? 2 : 1) * 3;        // resident waves, three rounds queued
    const dim3 grid((unsigned)(count < cap ? count : cap));
    const int dbg = wmf_debug_flags;
#ifdef WMF_LAB
    const bool x6 = !(dbg & 8192);                              // debug flag 8192 (lab builds only): f32 MFMA accumulation
#else
    constexpr bool x6 = true;                                   // (the f32-MFMA accumulation variants are compiled into -DWMF_LAB builds only)
#endif
#define DL_LAUNCH(N, B, X) WMF_LAUNCH(dl_name("solve_directl_kernel<" #N ", " #B ", " #X ", 16, 0>", count_dev != nullptr), (solve_directl_kernel<N, B, X>), grid, dim3(64), \
                                      DL_LDSB(N, 16), st, rows, count, V, side, indptr, indices, vals, f, ld, g, fb_rows, fb_count, dbg, \
                                      (const int64_t*)nullptr, (const int32_t*)nullptr, (float*)nullptr, count_dev)
#ifdef WMF_LAB
#define DL_PICK(N) do { if (f % 16) { if (x6) DL_LAUNCH(N, true, true); else DL_LAUNCH(N, true, false); } \
                        else        { if (x6) DL_LAUNCH(N, false, true); else DL_LAUNCH(N, false, false); } } while (0)
#else
#define DL_PICK(N) do { if (f % 16) DL_LAUNCH(N, true, true); else DL_LAUNCH(N, false, true); } while (0)
#endif
    if (nfb == 4) DL_PICK(4);
    else if (x6 && !(dbg & 16777216)) {
        // k = 128: 8-entry groups (16 KB ring, 16-entry chunks with 16x16x16 MFMAs, w_p in LDS by inline asm, lane coordinates
        // re-formed at every pivot), TWO waves per SIMD on 256 registers each (249 used, no scratch).  The second wave
        // overlaps what one wave cannot -- rocprofv3 --pmc at cfg3: VALU active 80 % + MFMA busy 38 % of the SIMD cycles,
        // SQ_WAIT_ANY 16 % of the wave cycles, where the one-wave kernel (debug flag 16777216) shows 63 + 19 and 19 that
        // add up -- for 19.9 against 21.2 ms.  On the way there: every scratch reload is a VMEM operation that returns in
        // order, i.e. behind every LDS-DMA of the ring (27.3 ms with reloads of spilled lane constants inside the chunk loop,
        // 21.0 with none there, 19.9 with none at all); hipcc puts s_waitcnt vmcnt(0) in front of every LDS access it can see.
        const dim3 grid2((unsigned)(count < 2 * cap ? count : 2 * cap));
        if (f % 16) WMF_LAUNCH(dl_name("solve_directl_kernel<8, true, true, 8, 0>", count_dev != nullptr), (solve_directl_kernel<8, true, true, 8>), grid2, dim3(64), DL_LDSB(8, 8), st,
                               rows, count, V, side, indptr, indices, vals, f, ld, g, fb_rows, fb_count, dbg, (const int64_t*)nullptr,
                               (const int32_t*)nullptr, (float*)nullptr, count_dev);
        else WMF_LAUNCH(dl_name("solve_directl_kernel<8, false, true, 8, 0>", count_dev != nullptr), (solve_directl_kernel<8, false, true, 8>), grid2, dim3(64), DL_LDSB(8, 8), st,
                        rows, count, V, side, indptr, indices, vals, f, ld, g, fb_rows, fb_count, dbg, (const int64_t*)nullptr,
                        (const int32_t*)nullptr, (float*)nullptr, count_dev);
    } else DL_PICK(8);
#undef DL_PICK
#undef DL_LAUNCH
    return 0;
}

// the segments of rows with more than WMF_HEAVY_T entries at k = 128 (+- biases): partial systems by the two-waves kernel
int wmf_launch_directl_segments(int64_t nseg, const float* V, const float* side, const int32_t* indices, const float* vals, int f,
                                int ld, const int64_t* seg_lo, const int32_t* seg_d, float* partial, hipStream_t st) {
    if (nseg <= 0) return 0;
    if (!((f == 128 && ld == 128) || (f == 129 && ld == 132))) return -1;
    const int64_t cap = 256 * 4 * 2 * 3;
    const dim3 grid((unsigned)(nseg < cap ? nseg : cap));
    const int dbg = wmf_debug_flags;
    if (f % 16) WMF_LAUNCH("solve_directl_kernel<8, true, true, 8, 1>", (solve_directl_kernel<8, true, true, 8, 1>), grid, dim3(64), DL_LDSB(8, 8), st,
                           (const int32_t*)nullptr, nseg, V, side, (const int64_t*)nullptr, indices, vals, f, ld, (float*)nullptr,
                           (int32_t*)nullptr, (int32_t*)nullptr, dbg, seg_lo, seg_d, partial, (const int32_t*)nullptr);
    else WMF_LAUNCH("solve_directl_kernel<8, false, true, 8, 1>", (solve_directl_kernel<8, false, true, 8, 1>), grid, dim3(64), DL_LDSB(8, 8), st,
                    (const int32_t*)nullptr, nseg, V, side, (const int64_t*)nullptr, indices, vals, f, ld, (float*)nullptr,
                    (int32_t*)nullptr, (int32_t*)nullptr, dbg, seg_lo, seg_d, partial, (const int32_t*)nullptr);
    return 0;
}
