// Stable LSD radix sort for gfx950, 4 bits per pass (wmf_sort.h).  Per pass three steps:
//   sort_hist_kernel     one workgroup per TILE of 2048 consecutive elements counts the tile's 16 digit values -> hist[digit][tile]
//   sort_scan*_kernel    exclusive prefix sum of hist in (digit, tile) order: where each tile's elements of each digit go
//   sort_scatter_kernel  the tile again: thread t holds 8 CONSECUTIVE elements; per-thread digit counts in LDS ([digit][thread]),
//                        one block-wide exclusive scan of those 4096 counters gives every (digit, thread) its first in-tile rank,
//                        and a thread numbers its own elements in order from there -- elements of equal digit leave the tile in
//                        input order, tiles follow each other in the global scan: the pass is stable, so the whole sort is.
// No atomics on the data path (the histogram's LDS counters aside), nothing depends on the launch: bit-reproducible.
// 16 buckets keep a tile's writes in runs of ~128 elements; a 44-bit key (10 M x 1 M entries) takes 11 passes of ~3 bytes
// moved per key byte -- setup work, once per train().
#include "wmf_sort.h"
#include "wmf_internal.h"

#define SORT_THREADS 256
#define SORT_ITEMS 8
#define SORT_TILE (SORT_THREADS * SORT_ITEMS)
#define SORT_SCAN_CHUNK 2048                      /* entries of hist one workgroup scans */

__global__ __launch_bounds__(SORT_THREADS) void sort_hist_kernel(const unsigned long long* __restrict__ keys, int64_t n, int shift,
                                                                 uint32_t* __restrict__ hist, int64_t ntiles) {
    __shared__ uint32_t cnt[16];
    const int t = threadIdx.x;
    if (t < 16) cnt[t] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * SORT_TILE + (int64_t)t * SORT_ITEMS;
    uint32_t mine[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) mine[d] = 0;
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        if (base + i < n) {
            const int d = (int)((keys[base + i] >> shift) & 15ull);
#pragma unroll
            for (int e = 0; e < 16; ++e) mine[e] += (e == d) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int d = 0; d < 16; ++d) if (mine[d]) atomicAdd(&cnt[d], mine[d]);
    __syncthreads();
    if (t < 16) hist[(int64_t)t * ntiles + blockIdx.x] = cnt[t];
}

// exclusive scan of `a` (len entries) in three steps: chunk-local scan + chunk totals, scan of the totals, add back
__device__ __forceinline__ uint32_t sort_block_exclusive(uint32_t v, uint32_t* sh, uint32_t* total) {
    const int t = threadIdx.x;                                   // Hillis-Steele over SORT_THREADS values in LDS
    sh[t] = v;
    __syncthreads();
#pragma unroll
    for (int o = 1; o < SORT_THREADS; o <<= 1) {
        const uint32_t add = t >= o ? sh[t - o] : 0u;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    const uint32_t incl = sh[t];
    if (total) *total = sh[SORT_THREADS - 1];
    __syncthreads();
    return incl - v;
}

__global__ __launch_bounds__(SORT_THREADS) void sort_scan1_kernel(uint32_t* __restrict__ a, int64_t len, uint32_t* __restrict__ sums) {
    __shared__ uint32_t sh[SORT_THREADS];
    const int t = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * SORT_SCAN_CHUNK + (int64_t)t * (SORT_SCAN_CHUNK / SORT_THREADS);
    uint32_t v[SORT_SCAN_CHUNK / SORT_THREADS], s = 0;
#pragma unroll
    for (int i = 0; i < SORT_SCAN_CHUNK / SORT_THREADS; ++i) { v[i] = base + i < len ? a[base + i] : 0u; s += v[i]; }
    uint32_t total;
    uint32_t run = sort_block_exclusive(s, sh, &total);
#pragma unroll
    for (int i = 0; i < SORT_SCAN_CHUNK / SORT_THREADS; ++i) { if (base + i < len) a[base + i] = run; run += v[i]; }
    if (t == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(SORT_THREADS) void sort_scan2_kernel(uint32_t* __restrict__ sums, int64_t nchunks) {
    __shared__ uint32_t sh[SORT_THREADS];
    __shared__ uint32_t carry_s;
    const int t = threadIdx.x;
    if (t == 0) carry_s = 0;
    __syncthreads();
    for (int64_t lo = 0; lo < nchunks; lo += SORT_THREADS) {       // one workgroup walks the chunk totals
        const uint32_t v = lo + t < nchunks ? sums[lo + t] : 0u;
        uint32_t total;
        const uint32_t ex = sort_block_exclusive(v, sh, &total);
        const uint32_t carry = carry_s;
        if (lo + t < nchunks) sums[lo + t] = carry + ex;
        __syncthreads();
        if (t == 0) carry_s = carry + total;
        __syncthreads();
    }
}

__global__ __launch_bounds__(SORT_THREADS) void sort_scan3_kernel(uint32_t* __restrict__ a, int64_t len, const uint32_t* __restrict__ sums) {
    const uint32_t add = sums[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * SORT_SCAN_CHUNK;
    for (int i = threadIdx.x; i < SORT_SCAN_CHUNK; i += SORT_THREADS) if (base + i < len) a[base + i] += add;
}

template <bool PAIRS>
__global__ __launch_bounds__(SORT_THREADS) void sort_scatter_kernel(const unsigned long long* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                                    unsigned long long* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                                    int64_t n, int shift, const uint32_t* __restrict__ hist, int64_t ntiles) {
    __shared__ uint32_t c[16 * SORT_THREADS];                    // [digit][thread]: counts, then first in-tile ranks
    __shared__ uint32_t sh[SORT_THREADS];
    __shared__ uint32_t dstart[16];                              // in-tile rank of the tile's first element of each digit
    const int t = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * SORT_TILE + (int64_t)t * SORT_ITEMS;
    unsigned long long k[SORT_ITEMS];
    uint32_t v[SORT_ITEMS];
    int dg[SORT_ITEMS];
#pragma unroll
    for (int d = 0; d < 16; ++d) c[d * SORT_THREADS + t] = 0;
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        const bool on = base + i < n;
        k[i] = on ? keys_in[base + i] : 0ull;
        if constexpr (PAIRS) v[i] = on ? vals_in[base + i] : 0u;
        dg[i] = on ? (int)((k[i] >> shift) & 15ull) : -1;
        if (on) c[dg[i] * SORT_THREADS + t] += 1u;               // this thread's own slots: no race
    }
    __syncthreads();
    // exclusive scan of the 4096 counters in (digit, thread) order: thread t takes the entries 16 t .. 16 t + 15
    uint32_t loc[16], s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) { loc[i] = c[16 * t + i]; s += loc[i]; }
    uint32_t run = sort_block_exclusive(s, sh, nullptr);
#pragma unroll
    for (int i = 0; i < 16; ++i) { c[16 * t + i] = run; run += loc[i]; }
    __syncthreads();
    if (t < 16) dstart[t] = c[t * SORT_THREADS];                 // (before thread 0 starts advancing its own counters)
    __syncthreads();
#pragma unroll
    for (int i = 0; i < SORT_ITEMS; ++i) {
        if (dg[i] < 0) continue;
        const int d = dg[i];
        const uint32_t rank = c[d * SORT_THREADS + t]++;         // in-tile rank in (digit, thread, item) order; own slot: no race
        const int64_t dst = (int64_t)hist[(int64_t)d * ntiles + blockIdx.x] + (int64_t)(rank - dstart[d]);
        keys_out[dst] = k[i];
        if constexpr (PAIRS) vals_out[dst] = v[i];
    }
}

size_t wmf_sort_ws_bytes(int64_t n) {
    if (n <= 0) return 256;
    const int64_t ntiles = (n + SORT_TILE - 1) / SORT_TILE, len = 16 * ntiles, nchunks = (len + SORT_SCAN_CHUNK - 1) / SORT_SCAN_CHUNK;
    return (size_t)((len + nchunks + 64) * 4 + 512);
}

int wmf_sort_u64(unsigned long long* keys, unsigned long long* keys_alt, uint32_t* vals, uint32_t* vals_alt, int64_t n, int bits,
                 void* ws, hipStream_t st, bool* in_alt) {
    *in_alt = false;
    if (n <= 1 || bits <= 0) return 0;
    if (n >= (1ll << 32)) return -4;                             // tile offsets and the payload are 32-bit (wmf_sort.h: "too many keys")
    const int64_t ntiles = (n + SORT_TILE - 1) / SORT_TILE, len = 16 * ntiles, nchunks = (len + SORT_SCAN_CHUNK - 1) / SORT_SCAN_CHUNK;
    uint32_t* hist = static_cast<uint32_t*>(ws);
    uint32_t* sums = hist + ((len + 63) / 64) * 64;
    unsigned long long *ki = keys, *ko = keys_alt;
    uint32_t *vi = vals, *vo = vals_alt;
    static const char* nm_h = wmf_kname("sort_hist_kernel");
    static const char* nm_s = wmf_kname("sort_scatter_kernel");
    for (int shift = 0; shift < bits; shift += 4) {
        WMF_LAUNCH(nm_h, sort_hist_kernel, dim3((unsigned)ntiles), dim3(SORT_THREADS), 0, st, ki, n, shift, hist, ntiles);
        hipLaunchKernelGGL(sort_scan1_kernel, dim3((unsigned)nchunks), dim3(SORT_THREADS), 0, st, hist, len, sums);
        hipLaunchKernelGGL(sort_scan2_kernel, dim3(1), dim3(SORT_THREADS), 0, st, sums, nchunks);
        hipLaunchKernelGGL(sort_scan3_kernel, dim3((unsigned)nchunks), dim3(SORT_THREADS), 0, st, hist, len, sums);
        if (vals) WMF_LAUNCH(nm_s, (sort_scatter_kernel<true>), dim3((unsigned)ntiles), dim3(SORT_THREADS), 0, st, ki, vi, ko, vo, n, shift, hist, ntiles);
        else WMF_LAUNCH(nm_s, (sort_scatter_kernel<false>), dim3((unsigned)ntiles), dim3(SORT_THREADS), 0, st, ki, vi, ko, vo, n, shift, hist, ntiles);
        unsigned long long* tk = ki; ki = ko; ko = tk;
        uint32_t* tv = vi; vi = vo; vo = tv;
        *in_alt = !*in_alt;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
