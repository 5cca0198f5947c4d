// Shared device helpers for the WMF/ALS kernels (gfx950 only: 64-wide waves, f32 MFMA, DPP).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <utility>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32: D[i][j] += sum_k A[i][k] * B[k][j], i,j in [0,16), k in [0,4).
//   lane l supplies A[l & 15][l >> 4] and B[l >> 4][l & 15];
//   lane l holds    D[4 * (l >> 4) + reg][l & 15] in acc[reg].
// Exact f32 (an fma chain in k order), same rate as the f32 VALU peak.
#define WMF_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int CTRL>
__device__ __forceinline__ float wmf_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// Sum over the 16 lanes of a DPP row (lanes 16g .. 16g+15); every lane of the row gets the total.
__device__ __forceinline__ float wmf_row16_sum(float v) {
    v += wmf_dpp<0xB1>(v);    // quad_perm [1,0,3,2]
    v += wmf_dpp<0x4E>(v);    // quad_perm [2,3,0,1]
    v += wmf_dpp<0x141>(v);   // row_half_mirror
    v += wmf_dpp<0x140>(v);   // row_mirror
    return v;
}

// Four independent row sums as 16 v_add_f32_dpp (hipcc leaves a v_mov_dpp + add pair per stage).  The four
// values are interleaved so that three instructions separate a register's write from its DPP read (the
// hazard needs two wait states); the leading s_nop covers the first stage.
__device__ __forceinline__ void wmf_row16_sum4(float& a, float& b, float& c, float& d) {
#define WMF_STAGE(ctrl)                                                  \
    "v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf\n\t"
    asm volatile("s_nop 1\n\t" WMF_STAGE("quad_perm:[1,0,3,2]") WMF_STAGE("quad_perm:[2,3,0,1]")
                 WMF_STAGE("row_half_mirror") WMF_STAGE("row_mirror")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef WMF_STAGE
}

// Four row sums for the price of less than three: each exchange stage also HALVES the number of values a lane carries
// (it keeps the half its partner sends it and sends the other one), so the stages cost 4 + 2, 2 + 1, 1, 1 instructions
// instead of 4 each.  On return `a` holds, on every lane (r, .), the 16-lane sum of input number 2 (r >> 3) + ((r >> 2) & 1)
// -- a for r = 0..3, b for 4..7, c for 8..11, d for 12..15; b, c, d are clobbered.
__device__ __forceinline__ void wmf_row16_sum4_scatter(float& a, float& b, float& c, float& d) {
    float t0, t1;
    asm volatile("s_nop 1\n\t"
                 "v_cndmask_b32 %4, %2, %0, %6\n\t"            // r >= 8 keeps (c, d) and sends (a, b); r < 8 the other way round
                 "v_cndmask_b32 %5, %3, %1, %6\n\t"
                 "v_cndmask_b32 %0, %0, %2, %6\n\t"
                 "v_cndmask_b32 %1, %1, %3, %6\n\t"
                 "v_add_f32_dpp %0, %4, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %5, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "v_cndmask_b32 %4, %1, %0, %7\n\t"            // r & 4 keeps the second of the two and sends the first
                 "v_cndmask_b32 %0, %0, %1, %7\n\t"
                 "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %4, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&v"(t0), "=&v"(t1)
                 : "s"(0xFF00FF00FF00FF00ull), "s"(0xF0F0F0F0F0F0F0F0ull));
}

// the same over the two 8-lane halves of each DPP row (lanes 16g .. 16g+7 and 16g+8 .. 16g+15) separately
__device__ __forceinline__ void wmf_row8_sum4(float& a, float& b, float& c, float& d) {
#define WMF_STAGE(ctrl)                                                  \
    "v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %2, %2, %2 " ctrl " row_mask:0xf bank_mask:0xf\n\t" \
    "v_add_f32_dpp %3, %3, %3 " ctrl " row_mask:0xf bank_mask:0xf\n\t"
    asm volatile("s_nop 1\n\t" WMF_STAGE("quad_perm:[1,0,3,2]") WMF_STAGE("quad_perm:[2,3,0,1]") WMF_STAGE("row_half_mirror")
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef WMF_STAGE
}

__device__ __forceinline__ double wmf_wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// d_i += nf * bcast(s_i), bcast = the value of lane KK of each 16-lane DPP row (row_newbcast, gfx90a+),
// as ONE v_fmac_f32_dpp per element (hipcc does not fold a DPP mov into the fma by itself).  The
// leading s_nop covers the "VALU write -> DPP read" hazard for the first source; no source of a later
// instruction is written by an earlier one in the block.
template <int KK>
__device__ __forceinline__ void fmac_bcast4(float& d0, float& d1, float& d2, float& d3, float s0, float s1, float s2,
                                            float s3, float nf) {
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f32_dpp %0, %4, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %1, %5, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %2, %6, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %3, %7, %8 row_newbcast:%9 row_mask:0xf bank_mask:0xf"
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)
                 : "v"(s0), "v"(s1), "v"(s2), "v"(s3), "v"(nf), "n"(KK));
}

// same with source == destination (every lane reads its row's lane KK before any lane is written)
template <int KK>
__device__ __forceinline__ void fmac_bcast4_self(float& d0, float& d1, float& d2, float& d3, float nf) {
    asm volatile("s_nop 1\n\t"
                 "v_fmac_f32_dpp %0, %0, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %1, %1, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %2, %2, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %3, %3, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
                 : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)
                 : "v"(nf), "n"(KK));
}

__device__ __forceinline__ float rlw(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// Value of 16-lane row group KQ (lanes 16 KQ .. 16 KQ + 15) copied to all four row groups, lane for lane,
// with two VALU swaps and no LDS round trip (v_permlane32_swap: lanes 32-63 of a <-> lanes 0-31 of b;
// v_permlane16_swap: odd rows of a <-> even rows of b).
template <int KQ>
__device__ __forceinline__ float wmf_bcast_rowgroup(float v) {
    const int x = __builtin_bit_cast(int, v);
    const auto s = __builtin_amdgcn_permlane32_swap(x, x, false, false);      // {g0 g1 g0 g1}, {g2 g3 g2 g3}
    const int h = (KQ < 2) ? s[0] : s[1];
    const auto t = __builtin_amdgcn_permlane16_swap(h, h, false, false);      // {ga ga ga ga}, {gb gb gb gb}
    return __builtin_bit_cast(float, (KQ & 1) ? t[1] : t[0]);
}

// ---- split-f16 operands: x = hi + lo with hi = RN_f16(x) and lo = RN_f16(x - hi) (22 significand bits).  Written out as
// instructions because hipcc takes the low part the long way round (v_cvt_f32_f16 of the high part, then a subtraction):
// v_fma_mix_f32 reads the packed f16 high part directly, so four values cost 2 + 4 + 2 instructions instead of 12.
// ONLY for values that come from loads or VALU arithmetic: hipcc does not look into an asm block, so it keeps neither the
// wait states an MFMA result needs before a VALU instruction may read it, nor those an MFMA needs behind a VALU write of
// its operand (the trailing s_nop below covers the second case; nothing cheap covers the first).  Bit-identical to the
// C++ split (tools/lab/split_probe.hip).
typedef unsigned wmf_u32x4 __attribute__((ext_vector_type(4)));
// { hi(x0) hi(x1) | hi(x2) hi(x3) | lo(x0) lo(x1) | lo(x2) lo(x3) }
__device__ __forceinline__ wmf_u32x4 wmf_split4(float x0, float x1, float x2, float x3) {
    unsigned h01, h23, l01, l23;
    float l0, l1, l2, l3;
    asm("v_cvt_pk_f16_f32 %0, %6, %7\n\t"
        "v_cvt_pk_f16_f32 %1, %8, %9\n\t"
        "v_fma_mix_f32 %2, %6, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %3, %7, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %4, %8, 1.0, -%1 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %5, %9, 1.0, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h01), "=&v"(h23), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
    // (the trailing s_nop: hipcc does not see what an asm block writes, so it cannot keep the two wait states an MFMA needs
    // behind a VALU write of one of its operands -- a consumer one instruction further down read stale registers)
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5\n\ts_nop 1" : "=&v"(l01), "=v"(l23) : "v"(l0), "v"(l1), "v"(l2), "v"(l3));
    return wmf_u32x4{h01, h23, l01, l23};
}
// The same for values that may be MFMA RESULTS: twelve wait states in front (an 8-pass MFMA's result may be read by a VALU
// instruction 11 cycles after the MFMA issued; hipcc inserts them for instructions it can see into, not for an asm block).
// Worth it where a second wave fills the wait: four VALU instructions fewer per split.
__device__ __forceinline__ wmf_u32x4 wmf_split4_after_mfma(float x0, float x1, float x2, float x3) {
    unsigned h01, h23, l01, l23;
    float l0, l1, l2, l3;
    asm volatile("s_nop 7\n\ts_nop 3\n\t"
        "v_cvt_pk_f16_f32 %0, %6, %7\n\t"
        "v_cvt_pk_f16_f32 %1, %8, %9\n\t"
        "v_fma_mix_f32 %2, %6, 1.0, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %3, %7, 1.0, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %4, %8, 1.0, -%1 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %5, %9, 1.0, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h01), "=&v"(h23), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5\n\ts_nop 1" : "=&v"(l01), "=v"(l23) : "v"(l0), "v"(l1), "v"(l2), "v"(l3));
    return wmf_u32x4{h01, h23, l01, l23};
}
// the same for the scaled values x_i s_i: hi = RN_f16(RN_f32(x s)), lo against the EXACT product (fma), so hi + lo is x s to
// 2^-22 whatever the rounding of the f32 product was
__device__ __forceinline__ wmf_u32x4 wmf_split4_scaled(float x0, float x1, float x2, float x3, float s0, float s1, float s2, float s3) {
    unsigned h01, h23, l01, l23;
    float l0, l1, l2, l3;
    asm("v_mul_f32 %2, %6, %10\n\t"
        "v_mul_f32 %3, %7, %11\n\t"
        "v_mul_f32 %4, %8, %12\n\t"
        "v_mul_f32 %5, %9, %13\n\t"
        "v_cvt_pk_f16_f32 %0, %2, %3\n\t"
        "v_cvt_pk_f16_f32 %1, %4, %5\n\t"
        "v_fma_mix_f32 %2, %6, %10, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %3, %7, %11, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %4, %8, %12, -%1 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mix_f32 %5, %9, %13, -%1 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h01), "=&v"(h23), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
        : "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(s0), "v"(s1), "v"(s2), "v"(s3));
    // (the trailing s_nop: hipcc does not see what an asm block writes, so it cannot keep the two wait states an MFMA needs
    // behind a VALU write of one of its operands -- a consumer one instruction further down read stale registers)
    asm("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5\n\ts_nop 1" : "=&v"(l01), "=v"(l23) : "v"(l0), "v"(l1), "v"(l2), "v"(l3));
    return wmf_u32x4{h01, h23, l01, l23};
}

// Sum of a value over the four 16-lane row groups, lane for lane, in every group: seven VALU instructions and no LDS round
// trip (two ds_bpermute -- __shfl_xor 16, 32 -- cost a wave that runs alone on its SIMD some 200 exposed cycles).
__device__ __forceinline__ float wmf_qsum(float v) {
    const int x = __builtin_bit_cast(int, v);
    const auto s = __builtin_amdgcn_permlane32_swap(x, x, false, false);      // {g0 g1 g0 g1}, {g2 g3 g2 g3}
    const float t = __builtin_bit_cast(float, (int)s[0]) + __builtin_bit_cast(float, (int)s[1]);   // {g0+g2, g1+g3, g0+g2, g1+g3}
    const int y = __builtin_bit_cast(int, t);
    const auto u = __builtin_amdgcn_permlane16_swap(y, y, false, false);      // {t0 t0 t0 t0}, {t1 t1 t1 t1}
    return __builtin_bit_cast(float, (int)u[0]) + __builtin_bit_cast(float, (int)u[1]);
}

// In-place Gauss-Jordan inverse of a symmetric positive definite 16 x 16 tile held row-distributed:
// lane (r, q) has A[r][4q + reg] in a[reg].  Step K: column K becomes e_K first, then every row i != K gets
// row_i += nf_i * row_K with nf_i = -A[i][K] / piv.  The pivot row itself is NOT normalised inside the sweep: rows that have
// been pivots stay at piv_K times their final value (the updates of the later steps are linear in the row, so the factor
// rides along), every lane remembers 1 / piv of its own row in `dsc`, and the caller's sweep multiplies once at the end,
// X = diag(dsc) . tile.  The obvious in-place form, row_K += (1 / piv - 1) row_K, rounds its multiplier at 2^-24 of ONE, not
// of 1 / piv: a relative error of 6e-8 piv in the whole row, i.e. wrong inverses for confidence weights far above the
// benchmark's (pre_process_count = 'linear' on large counts, RecModel/wmf_model.py:122-123; tests: test_weight_range_...).
// The multiplier A[r][K] sits in lane (r, K / 4); LDS = true fetches it with one ds_bpermute (one LDS instruction,
// long latency), false with two VALU lane swaps (five VALU instructions, no LDS round trip).
// CHECK = false skips the pivot test: for callers whose tile is positive definite by construction (pivots >= 1)
// and who test the result for NaN / Inf anyway.  CAP: a tile whose pivots spread over more than a factor WMF_PIVOT_SPREAD
// (or exceed WMF_PIVOT_CAP) also clears `ok` -- for the block eliminations that multiply by the explicit inverse of the pivot
// tile: that is as accurate as a triangular solve only while the tile is well conditioned (errors grow with cond(tile)^2;
// the pivot spread is the cheap proxy for cond(tile)), and the split-f16 parts of 1 / piv lose bits to the f16 subnormal
// range as piv grows.  Whitened systems under the usual confidence weights have tile pivots within a factor of two or three of
// each other -- 1.0 .. 1.3 on the benchmark matrices, 3 .. 11 in the 7 M-entry item rows of their Zipf variants -- so neither
// test fires there; a row that MIXES weights of 1e4 and more with ordinary ones (alpha * count, 'linear') goes to the pivoted LU
// kernel instead, which is slower and as accurate as float32 allows (measured on MI355X, tests/scale/diag_weights_mixed.py:
// 1e-3 at cond 1e4 where the tile-inverse path gave 0.1 .. 0.5).
#ifndef WMF_PIVOT_CAP
#define WMF_PIVOT_CAP 256.f
#endif
#ifndef WMF_PIVOT_SPREAD
#define WMF_PIVOT_SPREAD 8.f
#endif
template <int K, bool LDS, bool CHECK, bool CAP = false>
__device__ __forceinline__ void gj_inv_step(f32x4& a, float& dsc, const int (&baddr)[4], int r, int q, bool& ok, float& plo, float& phi) {
    constexpr int kq = K >> 2, kr = K & 3;
    const float akr = a[kr];     // copy first: __builtin_bit_cast applied to the vector-element lvalue itself reads element 0
    const float piv = rlw(akr, K + 16 * kq);
    if constexpr (CHECK) { if (!(piv > 1e-20f)) ok = false; }
    // (wave-uniform values: scalar arithmetic.  A pivot of exactly one belongs to a padding column -- f not a multiple of 16 -- or
    // to a feature none of the row's entries touches: an identity row and column, decoupled from the rest of the tile, which
    // says nothing about the conditioning of what it is decoupled from)
    if constexpr (CAP) { if (piv != 1.f) plo = fminf(plo, piv); phi = fmaxf(phi, piv); }
    const float inv = __builtin_amdgcn_rcpf(piv);
    float fk;
    if constexpr (LDS) fk = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(baddr[kq], __builtin_bit_cast(int, akr)));
    else               fk = wmf_bcast_rowgroup<kq>(akr);
    if (q == kq) a[kr] = (r == K) ? 1.f : 0.f;
    const float nf = (r == K) ? 0.f : -fk * inv;
    dsc = (r == K) ? inv : dsc;
    float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
    fmac_bcast4_self<K>(a0, a1, a2, a3, nf);
    a[0] = a0; a[1] = a1; a[2] = a2; a[3] = a3;
}
template <bool LDS = true, bool CHECK = true, bool CAP = false, int... Ks>
__device__ __forceinline__ void gj_inv_sweep(f32x4& a, const int (&baddr)[4], int r, int q, bool& ok,
                                             std::integer_sequence<int, Ks...>) {
    float dsc = 1.f, plo = 3.0e38f, phi = 0.f;
    (gj_inv_step<Ks, LDS, CHECK, CAP>(a, dsc, baddr, r, q, ok, plo, phi), ...);
    if constexpr (CAP) { if (phi > WMF_PIVOT_CAP || phi > WMF_PIVOT_SPREAD * plo) ok = false; }
    a *= dsc;
}

// The same step for a wave that runs alone on its SIMD (the heavy-row kernels): every instruction costs issue time there,
// so the lane masks are compile-time constants in scalar registers instead of compares (r == K: lanes K, K + 16, ..;
// q == K / 4: one 16-lane group), and the multiplier is formed as  nf = (e_K f - f) / piv,  e_K = [r == K]  -- which is
// 0 in the pivot row (un-normalised sweep, as above) and -f / piv elsewhere, the e_K doubling as the preset of column K.
// 15 VALU instructions a step where the step above compiles to 20.  No pivot test: the caller checks the result.
// BP: the multiplier column by ONE ds_bpermute_b32 (r4 = 4 (lane & 15), the row group in the instruction's offset field, the
// wait in the same asm block so that hipcc sees neither) instead of five VALU instructions -- for callers that run two
// waves per SIMD, where the permute's latency is the other wave's issue time.
template <int K, bool BP = false>
__device__ __forceinline__ void gj_inv_step_lean(f32x4& a, float& dsc, int& pmin, int& pmax, int r4 = 0) {
    constexpr int kq = K >> 2, kr = K & 3;
    const float akr = a[kr];
    const float piv = rlw(akr, K + 16 * kq);
    // pivot tests on the scalar unit: the smallest and the largest pivot BIT PATTERN as signed integers -- for positive floats
    // the integer order is the float order, a negative pivot is a negative integer -- which the caller compares with those of
    // 1e-20f (WMF_PIVOT_MIN_BITS) and of WMF_PIVOT_CAP once.  A NaN passes, and reaches the solution, which the caller tests.
    // (a pivot of exactly one -- padding column, untouched feature: an identity row and column -- is left out of the minimum:
    // it is positive, and decoupled from the tile whose conditioning the spread stands for)
    const int pbits = __builtin_bit_cast(int, piv);
    pmin = pbits == 0x3f800000 ? pmin : min(pmin, pbits);
    pmax = max(pmax, pbits);
    const float inv = __builtin_amdgcn_rcpf(piv);
    float fk;
    if constexpr (BP) asm volatile("ds_bpermute_b32 %0, %1, %2 offset:%3\n\ts_waitcnt lgkmcnt(0)" : "=v"(fk) : "v"(r4), "v"(akr), "n"(64 * kq));
    else fk = wmf_bcast_rowgroup<kq>(akr);
    // (the masks are shifted into place next to their use: as C++ constants hipcc computes all twenty once, ahead of the
    // eight sweeps of a row, and then spills them to VGPR lanes -- v_writelane / v_readlane around every use)
    // eK = [r == K]; dsc = 1 / piv on the lanes of row K (the un-normalised sweep, see gj_inv_step); pre = column K preset
    float eK, pre;
    unsigned long long tmp;
    asm volatile("s_lshl_b64 %2, %5, %6\n\t"
                 "v_cndmask_b32 %0, 0, 1.0, %2\n\t"
#ifndef WMF_GJ_NORMALISED
                 "v_cndmask_b32 %3, %3, %8, %2\n\t"
#endif
                 "s_lshl_b64 %2, 0xffff, %7\n\t"
                 "v_cndmask_b32 %1, %4, %0, %2"
                 : "=&v"(eK), "=v"(pre), "=&s"(tmp), "+v"(dsc)
                 : "v"(akr), "s"(0x0001000100010001ull), "n"(K), "n"(16 * kq), "v"(inv)
                 : "scc");                                           // s_lshl_b64 writes SCC
    a[kr] = pre;
#ifdef WMF_GJ_NORMALISED                                             // lab only (tools/build_variant.sh): the round-2 step, for A/B timing
    const float nf = (eK - fk) * inv;
#else
    const float nf = __builtin_fmaf(eK, fk, -fk) * inv;              // 0 in the pivot row, -f / piv elsewhere
#endif
    float a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3];
    fmac_bcast4_self<K>(a0, a1, a2, a3, nf);
    a[0] = a0; a[1] = a1; a[2] = a2; a[3] = a3;
}
// (the caller multiplies the swept tile by dsc: X = diag(dsc) . a)
template <bool BP = false, int... Ks>
__device__ __forceinline__ void gj_inv_sweep_lean(f32x4& a, float& dsc, int& pmin, int& pmax, int& spread, int r4, std::integer_sequence<int, Ks...>) {
    dsc = 1.f;
    int tlo = 0x7f800000, thi = 0;                                   // this tile's smallest / largest pivot (bit patterns)
    (gj_inv_step_lean<Ks, BP>(a, dsc, tlo, thi, r4), ...);
    pmin = min(pmin, tlo);
    pmax = max(pmax, thi);
    spread = max(spread, thi - tlo);                                 // positive floats: the difference of the bit patterns ~ 2^23 log2(ratio)
}
#define WMF_PIVOT_MIN_BITS 0x1e3ce508
#define WMF_PIVOT_CAP_BITS 0x43800000          /* 256.f = WMF_PIVOT_CAP */
#define WMF_PIVOT_SPREAD_BITS (3 << 23)         /* a factor 8 = WMF_PIVOT_SPREAD between a tile's pivots */

// The same inverse as a SYMMETRIC sweep whose rank-one update is one f32 MFMA -- for callers that run two waves per SIMD and
// are bound by VALU issue, with MFMA time to spare.  The tile stays symmetric, so the row-distributed layout above is also
// the accumulator layout of v_mfma_f32_16x16x4_f32 (lane (r, q), register e: element [4q + e][r]), and column K -- lane
// (i, K / 4), register K & 3 -- is where the instruction reads A[i][k = K / 4] and B[k = K / 4][j] from.  Step K:
//     u = column K with u[K] = piv - 1,   v = -u / piv  (so v[K] = 1 / piv - 1),   A += u v^T,   A[K][K] = -1 / piv
// which is  A[i][j] -= A[i][K] A[K][j] / piv  off row and column K and scales those two by 1 / piv (the u[K], v[K] terms):
// the sweep operator; after the sixteen steps the tile holds MINUS the inverse.  v is written into a register that is zero
// outside lane group K / 4 (DPP row mask), which keeps the other three k slots of the instruction out of the sum.
// 7 VALU instructions and one MFMA a step instead of 14.  Same pivot test and caveats as the lean step.
template <int K>
__device__ __forceinline__ void gj_sweep_step_mfma(f32x4& a, int& pmin, float (&vz)[4]) {
    constexpr int kq = K >> 2, kr = K & 3;
    const float akr = a[kr];
    const float piv = rlw(akr, K + 16 * kq);
    pmin = min(pmin, __builtin_bit_cast(int, piv));
    const float inv = __builtin_amdgcn_rcpf(piv);
    float eK;                                                    // 1 in lane (K, K / 4), 0 elsewhere
    unsigned long long tmp;
    asm volatile("s_lshl_b64 %1, 1, %2\n\tv_cndmask_b32 %0, 0, 1.0, %1" : "=v"(eK), "=&s"(tmp) : "n"(K + 16 * kq) : "scc");
    const float u = akr - eK;
    // (two wait states between the VALU write of u and the DPP read, two between the write of v and the MFMA: hipcc keeps
    // neither around an asm block)
    asm volatile("s_nop 1\n\tv_mul_f32_dpp %0, %1, -%2 quad_perm:[0,1,2,3] row_mask:%3 bank_mask:0xf\n\ts_nop 1"
                 : "+v"(vz[kq]) : "v"(u), "v"(inv), "n"(1 << kq));
    a = WMF_MFMA16(u, vz[kq], a);
    a[kr] = (eK != 0.f) ? -inv : a[kr];
}
template <int... Ks>
__device__ __forceinline__ void gj_sweep_mfma(f32x4& a, int& pmin, std::integer_sequence<int, Ks...>) {
    float vz[4] = {0.f, 0.f, 0.f, 0.f};
    (gj_sweep_step_mfma<Ks>(a, pmin, vz), ...);
}

// value of lane (r, r >> 2) -- the same lane position in row group r >> 2 -- on every lane (r, q); hi8 / hi4: the lane masks
// r >= 8, (r & 4) != 0 as 64-bit constants
__device__ __forceinline__ float wmf_fetch_own_group(float v) {
    const int x = __builtin_bit_cast(int, v);
    const auto s = __builtin_amdgcn_permlane32_swap(x, x, false, false);      // {g0 g1 g0 g1}, {g2 g3 g2 g3}
    int h;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(h) : "v"((int)s[0]), "v"((int)s[1]), "s"(0xFF00FF00FF00FF00ull));   // r >= 8: groups 2, 3
    const auto t = __builtin_amdgcn_permlane16_swap(h, h, false, false);      // {lower group x 4}, {upper group x 4}
    int o;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(o) : "v"((int)t[0]), "v"((int)t[1]), "s"(0xF0F0F0F0F0F0F0F0ull));   // r & 4: the upper one
    return __builtin_bit_cast(float, o);
}

static inline int wmf_ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
