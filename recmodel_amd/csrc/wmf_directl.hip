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
//             slot[g % 4] + 1024 i: lanes 0..31 row A, lanes 32..63 row B, so that no instruction's kilobyte crosses a
//             row.  Ring position P = 2 i + h holds CSR entry 8 h + i of the group (h = lane >> 5): a lane then needs the
//             indices of eight CONSECUTIVE entries, two ds_read_b128.
//   consume   k-step t takes ring positions 4 t + q; lane (r, q) reads pieces r and r + 16 of its row (the feature
//             permutation of wmf_stream.h: virtual block 4 j + e = feature 64 j + 4 r + e), its weight and its border value.
// Everything hipcc must not see is inline asm: while an LDS-DMA is in flight the compiler waits vmcnt(0) before any LDS
// read or use of an ordinary load it knows of, which would drain the ring.  The waits are counted by hand (vmcnt is in
// issue order): group g has landed when at most 8 x (groups issued after it) vector-memory operations are outstanding.
// The elimination keeps w_p in registers (no LDS vectors), so no compiler-visible LDS access is left in the kernel.
#include "wmf_common.h"
#include "wmf_internal.h"
#include "wmf_dw_elim.h"

#include <utility>

#define DL_NFB 8
#define DL_R 4                      // ring slots = groups per 64-entry block (the group loop is unrolled by it)
#define DL_SLOT 8192                // bytes: 16 rows x 512
#define DL_META (DL_R * DL_SLOT)    // idx[4][64], w[4][64], border[4][64] behind the ring: metadata of block b in buffer b & 3
#define DL_W 1024                   // (block b + 1 is requested while groups of b - 1 are still being consumed: three live blocks)
#define DL_BD 2048
#define DL_LDS (DL_META + 3 * 1024)

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
template <int N>
__device__ __forceinline__ void dl_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <bool BORDER>
__global__ __launch_bounds__(64, 1) void solve_directl_kernel(const int32_t* __restrict__ rows, int64_t count, const float* __restrict__ V,
                                                              const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                              const float* __restrict__ vals, int f, int ld, float* __restrict__ g,
                                                              int32_t* __restrict__ fb_rows, int32_t* __restrict__ fb_count, int dbg) {
    constexpr int NFB = DL_NFB, NT = NFB * (NFB + 1) / 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int r = lane & 15, q = lane >> 4, h = lane >> 5;
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);   // LDS byte address of the region
    int baddr[4];
#pragma unroll
    for (int kq = 0; kq < 4; ++kq) baddr[kq] = (r + 16 * kq) * 4;
    // LDS byte addresses of this lane's reads (inline asm below)
    const unsigned ring_rd = lds0 + q * 512 + r * 16;                              // + slot * 8192 + t * 2048 + j * 256
    const unsigned meta_rd = lds0 + DL_META + (8 * (q & 1) + (q >> 1)) * 4;       // + (block & 3) * 256 + (group in block * 16 + 2 t) * 4
    const unsigned idx8_rd = lds0 + DL_META + 8 * h * 4;                          // + (block & 3) * 256 + group in block * 64
    const int piece = (lane & 31) * 4;                                      // first float of this lane's piece

    auto item = [&](int64_t i, int& u_, int64_t& lo_, int& d_) {
        u_ = rows[i]; lo_ = indptr[u_]; d_ = (int)(indptr[u_ + 1] - lo_);
    };
    // metadata of block b of row (lo_, d_): lane l <- entry min(64 b + l, d_ - 1) (clamped entries get weight 0 at use)
    auto issue_meta = [&](int b, int64_t lo_, int d_) {
        const int64_t e = lo_ + min(64 * b + lane, d_ - 1);
        const unsigned par = (b & 3) * 256;
        __builtin_amdgcn_global_load_lds((dl_gptr)(indices + e), (dl_lptr)(smem + DL_META + par), 4, 0, 0);
        __builtin_amdgcn_global_load_lds((dl_gptr)(vals + e), (dl_lptr)(smem + DL_META + DL_W + par), 4, 0, 0);
    };
    // border feature of the entries of block b (its indices have landed)
    auto issue_border = [&](int b) {
        if constexpr (BORDER) {
            const unsigned par = (b & 3) * 256;
            float iv = dl_read32<0>(lds0 + DL_META + par + lane * 4);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(iv)::"memory");     // the value passes through the wait: no use can move above it
            const int idx = __builtin_bit_cast(int, iv);
            __builtin_amdgcn_global_load_lds((dl_gptr)(V + (int64_t)idx * ld + 128), (dl_lptr)(smem + DL_META + DL_BD + par), 4, 0, 0);
        }
    };
    // the 16 rows of group gi (S = gi % 4 = its ring slot and its position in the block) -> 8 DMA instructions
    auto issue_rows = [&](auto slot, int gi) {
        constexpr int S = decltype(slot)::value;
        const unsigned par = ((gi >> 2) & 3) * 256;
        f32x4 ia = dl_read128<S * 64>(idx8_rd + par), ib = dl_read128<S * 64 + 16>(idx8_rd + par);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ia), "+v"(ib)::"memory");
        const float ids[8] = {ia[0], ia[1], ia[2], ia[3], ib[0], ib[1], ib[2], ib[3]};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int idx = __builtin_bit_cast(int, ids[i]);
            __builtin_amdgcn_global_load_lds((dl_gptr)(V + (int64_t)idx * ld + piece), (dl_lptr)(smem + S * DL_SLOT + i * 1024), 16, 0, 0);
        }
    };

    int u = 0, d = 0;
    int64_t lo = 0;
    int64_t it = blockIdx.x;
    // first requests of a row: metadata of block 0 (and 1), border of block 0, groups 0 .. 2.  Nothing else of this wave
    // is in flight when this runs, so the vmcnt(0) in it waits for the row's own first metadata only.
    auto prime = [&](int64_t lo_, int d_) {
        const int ng = (d_ + 15) >> 4;
        issue_meta(0, lo_, d_);
        dl_wait_vm<0>();
        issue_border(0);
        if (d_ > 64) issue_meta(1, lo_, d_);
        issue_rows(std::integral_constant<int, 0>{}, 0);
        if (ng > 1) issue_rows(std::integral_constant<int, 1>{}, 1);
        if (ng > 2) issue_rows(std::integral_constant<int, 2>{}, 2);
    };
    if (it < count) { item(it, u, lo, d); prime(lo, d); }

    for (; it < count; it += gridDim.x) {
        const int ngroups = (d + 15) >> 4;
        const int64_t itn = it + gridDim.x;
        int un = 0, dn = 0;
        int64_t lon = 0;
        if (itn < count) item(itn, un, lon, dn);       // used in prime() below, when nothing is in flight any more

        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
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
                    dl_wait_vm<32>();
                    issue_border(gn >> 2);
                    if (64 * ((gn >> 2) + 1) < d) issue_meta((gn >> 2) + 1, lo, d);
                }
                issue_rows(std::integral_constant<int, SN>{}, gn);
            }
            // group G has landed when only the requests made after it are outstanding
            const int younger = min(3, ngroups - 1 - G);
            if (younger >= 3) dl_wait_vm<24>();
            else if (younger == 2) dl_wait_vm<16>();
            else if (younger == 1) dl_wait_vm<8>();
            else dl_wait_vm<0>();
            if (dbg & 2) return;
            const unsigned par = ((G >> 2) & 3) * 256;
            auto kstep = [&](auto tc) {
                constexpr int t = decltype(tc)::value;
                f32x4 xa = dl_read128<S * DL_SLOT + t * 2048>(ring_rd), xb = dl_read128<S * DL_SLOT + t * 2048 + 256>(ring_rd);
                float wv = dl_read32<DL_W + (S * 16 + 2 * t) * 4>(meta_rd + par);
                float bf = 0.f;
                if constexpr (BORDER) bf = dl_read32<DL_BD + (S * 16 + 2 * t) * 4>(meta_rd + par);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xa), "+v"(xb), "+v"(wv), "+v"(bf)::"memory");
                // entry of ring position 4 t + q: 8 (P & 1) + (P >> 1), P & 1 = q & 1, P >> 1 = 2 t + (q >> 1)
                const bool real = 16 * G + 8 * (q & 1) + 2 * t + (q >> 1) < d;
                const float w = real ? wv : 0.f, p = real ? wv + 1.f : 0.f;
                const float x[NFB] = {xa[0], xa[1], xa[2], xa[3], xb[0], xb[1], xb[2], xb[3]};
                float fw[NFB];
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) { fw[fb] = x[fb] * w; racc[fb] += x[fb] * p; }
                if constexpr (BORDER) {
                    const float bw = bf * w;
#pragma unroll
                    for (int fb = 0; fb < NFB; ++fb) bacc[fb] += x[fb] * bw;
                    cacc += bf * bw;
                    eacc += bf * p;
                }
                int tt = 0;
#pragma unroll
                for (int bi = 0; bi < NFB; ++bi) {
#pragma unroll
                    for (int bj = bi; bj < NFB; ++bj, ++tt) acc[tt] = WMF_MFMA16(x[bi], fw[bj], acc[tt]);
                }
            };
            [&]<int... Ts>(std::integer_sequence<int, Ts...>) {
                (kstep(std::integral_constant<int, Ts>{}), ...);
            }(std::make_integer_sequence<int, 4>{});
        };
        for (int G0 = 0; G0 < ngroups; G0 += DL_R) {
            [&]<int... Ss>(std::integer_sequence<int, Ss...>) {
                (step(std::integral_constant<int, Ss>{}, G0 + Ss), ...);
            }(std::make_integer_sequence<int, DL_R>{});
        }
        if (itn < count) prime(lon, dn);                 // the next row's first 48 entries fly during the elimination

        // ---- C, D: block elimination and backward pass (wmf_dw_elim.h), w_p in registers
        bool ok = true;
        float gb[NFB];
        float tb = 0.f;
        dw_eliminate<NFB, BORDER, false, true>(acc, racc, bacc, cacc, eacc, nullptr, nullptr, r, q, baddr, dbg, gb, tb, ok);
        if (!ok) {
            if (lane == 0) fb_rows[atomicAdd(fb_count, 1)] = u;
        } else if (q == 0) {
#pragma unroll
            for (int p = 0; p < NFB; ++p) {
                const int c = 64 * (p >> 2) + 4 * r + (p & 3);    // undo the feature permutation
                g[(int64_t)u * ld + c] = gb[p];
            }
            if constexpr (BORDER) {
                const int c = 128 + r;                            // the border column and the padding behind it
                if (c < ld) g[(int64_t)u * ld + c] = (r == 0) ? tb : 0.f;
            }
        }
        u = un; lo = lon; d = dn;
    }
}

int wmf_directl_supported(int f, int ld) { return (f == 128 && ld == 128) || (f == 129 && ld == 132); }

int wmf_launch_directl(const int32_t* rows, int64_t count, const float* V, const int64_t* indptr, const int32_t* indices,
                       const float* vals, int f, int ld, float* g, int32_t* fb_rows, int32_t* fb_count, hipStream_t st) {
    if (count <= 0) return 0;
    if (!wmf_directl_supported(f, ld)) return -1;
    const int64_t cap = 256 * 4 * 3;                             // resident waves (one per SIMD), three rounds queued
    const dim3 grid((unsigned)(count < cap ? count : cap));
    const int dbg = wmf_debug_flags;
    if (f == 129)
        hipLaunchKernelGGL((solve_directl_kernel<true>), grid, dim3(64), DL_LDS, st, rows, count, V, indptr, indices, vals, f, ld, g,
                           fb_rows, fb_count, dbg);
    else
        hipLaunchKernelGGL((solve_directl_kernel<false>), grid, dim3(64), DL_LDS, st, rows, count, V, indptr, indices, vals, f, ld, g,
                           fb_rows, fb_count, dbg);
    return 0;
}
