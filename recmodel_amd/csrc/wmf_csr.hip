// COO -> CSR on the device (gfx950): the transpose of the confidence matrix (RecModel/wmf_model.py:128, `count_mat.T.tocsr()`)
// and every other re-sorting of stored entries the sharded engine needs (entries received from other ranks, the reduce
// mode's transposed shard).  Entries are sorted by (row, column) with a STABLE radix sort, so duplicates survive in their
// stored order -- the reference sums them implicitly because every stored entry is one term of the row's normal
// equations -- and the result does not depend on launch geometry.
//   keys_kernel      key = row * n_cols + col (64 bit), id = entry number
//   wmf_sort_u64     this library's own stable LSD radix sort (wmf_sort.hip), over the significant bits of the key only
//   gather_kernel    indices[e] = col of the e-th sorted entry (int32), values[e] = its value
//   indptr_kernel    row pointer from the sorted keys by boundary detection: entry e writes indptr[r] = e for every row r
//                    in (row of entry e - 1, row of entry e]; no histogram, no scan, no atomics
#include "wmf_internal.h"
#include "wmf_sort.h"

__global__ __launch_bounds__(256) void csr_keys_kernel(const int64_t* __restrict__ rows, const int64_t* __restrict__ cols,
                                                       int64_t nnz, int64_t n_rows, int64_t n_cols,
                                                       unsigned long long* __restrict__ keys, uint32_t* __restrict__ ids,
                                                       int32_t* __restrict__ bad) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * 256) {
        const int64_t r = rows[e], c = cols[e];
        if (r < 0 || r >= n_rows || c < 0 || c >= n_cols) { atomicExch(bad, 1); keys[e] = ~0ull; }
        else keys[e] = (unsigned long long)r * (unsigned long long)n_cols + (unsigned long long)c;
        ids[e] = (uint32_t)e;
    }
}

__global__ __launch_bounds__(256) void csr_gather_kernel(const unsigned long long* __restrict__ skeys, const uint32_t* __restrict__ sids,
                                                         const float* __restrict__ vals, int64_t nnz, int64_t n_cols,
                                                         int32_t* __restrict__ indices, float* __restrict__ values) {
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * 256) {
        indices[e] = (int32_t)(skeys[e] % (unsigned long long)n_cols);
        values[e] = vals[sids[e]];
    }
}

__global__ __launch_bounds__(256) void csr_indptr_kernel(const unsigned long long* __restrict__ skeys, int64_t nnz, int64_t n_rows,
                                                         int64_t n_cols, int64_t* __restrict__ indptr) {
    // position e = 0 .. nnz (one past the end closes the last rows)
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e <= nnz; e += (int64_t)gridDim.x * 256) {
        // (clamped: an out-of-range entry -- flagged by csr_keys_kernel, the call fails -- must not steer a store)
        const unsigned long long top = (unsigned long long)n_rows;
        const int64_t prev = e > 0 ? (int64_t)min(skeys[e - 1] / (unsigned long long)n_cols, top) : -1;
        const int64_t cur = e < nnz ? (int64_t)min(skeys[e] / (unsigned long long)n_cols, top) : n_rows;
        for (int64_t r = prev + 1; r <= cur; ++r) indptr[r] = e;          // (rows without entries: a run of equal pointers)
    }
}

static int key_bits(int64_t n_rows, int64_t n_cols) {
    const unsigned long long top = (unsigned long long)n_rows * (unsigned long long)n_cols;   // keys are < top
    int b = 1;
    while (b < 64 && (top >> b) != 0) ++b;
    return b;
}

static size_t al256(size_t b) { return (b + 255) / 256 * 256; }

// workspace: [keys nnz x 8][sorted keys][ids nnz x 4][sorted ids][flag 256][histograms of the sort]
int64_t wmf_csr_ws_bytes(int64_t nnz, int64_t n_rows, int64_t n_cols) {
    if (nnz <= 0) return 256;
    return (int64_t)(2 * al256((size_t)nnz * 8) + 2 * al256((size_t)nnz * 4) + 256 + wmf_sort_ws_bytes(nnz) + 256);
}

// returns 0, -2 (HIP failure), -3 (workspace too small), -4 (n_rows * n_cols does not fit 63 bits or nnz >= 2^32)
int wmf_launch_coo_to_csr(const int64_t* rows, const int64_t* cols, const float* vals, int64_t nnz, int64_t n_rows, int64_t n_cols,
                          int64_t* indptr, int32_t* indices, float* values, int32_t* bad_flag, void* ws, int64_t ws_bytes,
                          hipStream_t st) {
    if (n_rows > 0 && n_cols > 0 && (unsigned long long)n_rows > (0x7fffffffffffffffull / (unsigned long long)n_cols)) return -4;
    if (nnz >= (1ll << 32)) return -4;
    if (nnz <= 0) return hipMemsetAsync(indptr, 0, (size_t)(n_rows + 1) * 8, st) == hipSuccess ? 0 : -2;
    if (ws_bytes < wmf_csr_ws_bytes(nnz, n_rows, n_cols)) return -3;
    char* base = static_cast<char*>(ws);
    const size_t a8 = al256((size_t)nnz * 8), a4 = al256((size_t)nnz * 4);
    auto* keys = reinterpret_cast<unsigned long long*>(base);
    auto* skeys = reinterpret_cast<unsigned long long*>(base + a8);
    auto* ids = reinterpret_cast<uint32_t*>(base + 2 * a8);
    auto* sids = reinterpret_cast<uint32_t*>(base + 2 * a8 + a4);
    void* temp = base + 2 * a8 + 2 * a4 + 256;
    const int bits = key_bits(n_rows, n_cols);
    int64_t grid = (nnz + 255) / 256;
    if (grid > 16384) grid = 16384;
    WMF_LAUNCH("csr_keys_kernel", csr_keys_kernel, dim3((unsigned)grid), dim3(256), 0, st, rows, cols, nnz, n_rows, n_cols, keys, ids,
               bad_flag);
    bool in_alt = false;
    if (const int src = wmf_sort_u64(keys, skeys, ids, sids, nnz, bits, temp, st, &in_alt)) return src;      // (-2 launch failure, -4 too many keys)
    if (!in_alt) { auto* tk = keys; keys = skeys; skeys = tk; auto* ti = ids; ids = sids; sids = ti; }   // (skeys / sids: the sorted arrays)
    WMF_LAUNCH("csr_gather_kernel", csr_gather_kernel, dim3((unsigned)grid), dim3(256), 0, st, skeys, sids, vals, nnz, n_cols, indices,
               values);
    WMF_LAUNCH("csr_indptr_kernel", csr_indptr_kernel, dim3((unsigned)grid), dim3(256), 0, st, skeys, nnz, n_rows, n_cols, indptr);
    return 0;
}
