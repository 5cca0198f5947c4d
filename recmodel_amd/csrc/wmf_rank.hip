// Ranking-side kernels (gfx950), SURVEY.md section 8(f) row 4:
//   * hit_kernel   -- RecModel.eval_topn / compute_hit (RecModel/base_model.py:51-148): for every test entry
//                     (user, item) the reference ranks the item among `rand_sampled` random candidates with
//                     WMF.rank and asks whether it made the top n.  The item makes the top n exactly when fewer
//                     than n candidates score higher, so one wave per test entry scores the candidates (16 lanes
//                     per candidate, four at a time) and counts; hits are integer atomics, hence deterministic.
//                     The random draws stay on the host, in the reference's np.random call order.
//   * rank         -- WMF.rank (RecModel/wmf_model.py:25-47): scores of one user against a candidate list (the same
//                     predict kernel as eval_prec), then a top-n SELECT -- histogram, threshold bin, compaction -- and a
//                     sort of the short list that survives it (wmf_sort_u64, on a few thousandths of the
//                     candidates); the batched form (many users) sorts its score matrix by segments.

#include "wmf_sort.h"

#include "wmf_common.h"
#include "wmf_internal.h"

// same arithmetic as pair_score() in wmf_eval.hip (kept in step with it: eval_prec, predict and rank must agree)
__device__ __forceinline__ float rank_pair_score(const float* __restrict__ xu, const float* __restrict__ yi, int nch, int gl,
                                                 int bias) {
    float s = 0.f;
    for (int c = gl; c < nch; c += 16) {
        const float4 a = reinterpret_cast<const float4*>(xu)[c];
        const float4 b = reinterpret_cast<const float4*>(yi)[c];
        float first = a.x * b.x;
        if (bias && c == 0) first = a.x + b.x;
        s += first + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    return wmf_row16_sum(s);
}

__global__ __launch_bounds__(256) void hit_kernel(const float* __restrict__ users, const float* __restrict__ items, int ld,
                                                  int bias, const int32_t* __restrict__ pair_user,
                                                  const int32_t* __restrict__ pair_item, const int32_t* __restrict__ pair_row,
                                                  int64_t n_pairs, const int32_t* __restrict__ cand, int n_cand,
                                                  const int32_t* __restrict__ slot, const int32_t* __restrict__ topn, int n_topn,
                                                  unsigned long long* __restrict__ hits) {
    const int lane = threadIdx.x & 63, gl = lane & 15, grp = lane >> 4;
    const int nch = ld >> 2;
    for (int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); p < n_pairs; p += (int64_t)gridDim.x * 4) {
        const float* xu = users + (int64_t)pair_user[p] * ld;
        const float s_true = rank_pair_score(xu, items + (int64_t)pair_item[p] * ld, nch, gl, bias);
        const int32_t* crow = cand + (int64_t)pair_row[p] * n_cand;
        const int sl = slot[pair_row[p]];                          // this position holds the test item itself
        int higher = 0;
        for (int jb = 0; jb < n_cand; jb += 4) {                   // uniform trip count across the four groups
            const int j = jb + grp;
            const bool act = j < n_cand && j != sl;
            const float s = rank_pair_score(xu, items + (int64_t)crow[j < n_cand ? j : 0] * ld, nch, gl, bias);
            higher += (act && s > s_true) ? 1 : 0;
        }
        higher += __shfl_xor(higher, 16);
        higher += __shfl_xor(higher, 32);
        if (lane < n_topn && higher < topn[lane]) atomicAdd(&hits[lane], 1ULL);
    }
}

// ---- batched rank: many users against one candidate list ------------------------------------------------
// scores[u][c] for 16 users x 16 candidates per wave by f32 MFMA: both operands are row gathers of 16-byte pieces
// (lane (r, q) reads piece 4 kk + q of user r and of candidate r), exactly the S = V V^T pattern of wmf_solve.hip.
// bias: the column-0 product is left out and users[u][0] + items[c][0] added (wmf_model.py:209-211).
__global__ __launch_bounds__(256) void score_tile_kernel(const float* __restrict__ users, const float* __restrict__ items, int ld,
                                                         int bias, const int32_t* __restrict__ user_idx, int64_t nu,
                                                         const int32_t* __restrict__ cand_idx, int64_t nc,
                                                         float* __restrict__ scores) {
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const int nch = ld >> 2;
    const int64_t tiles_c = (nc + 15) >> 4, tiles = ((nu + 15) >> 4) * tiles_c;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); tile < tiles; tile += (int64_t)gridDim.x * 4) {
        const int64_t tu = tile / tiles_c, tc = tile % tiles_c;
        const int64_t uu = min(16 * tu + r, nu - 1), cc = min(16 * tc + r, nc - 1);          // clamped: stores are masked
        const float4* urow = reinterpret_cast<const float4*>(users + (int64_t)user_idx[uu] * ld);
        const float4* irow = reinterpret_cast<const float4*>(items + (int64_t)cand_idx[cc] * ld);
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
        float ub = 0.f, ib = 0.f;
        for (int c = q; c < ((nch + 3) & ~3); c += 4) {           // uniform trip count; pieces past the row are zero
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (c < nch) { a = urow[c]; b = irow[c]; }
            if (bias && c == 0) { ub = a.x; ib = b.x; a.x = 0.f; }
            acc = WMF_MFMA16(a.x, b.x, acc); acc = WMF_MFMA16(a.y, b.y, acc);
            acc = WMF_MFMA16(a.z, b.z, acc); acc = WMF_MFMA16(a.w, b.w, acc);
        }
        // acc[reg] = score(user 16 tu + 4 q + reg, candidate 16 tc + r); the biases sit in the q = 0 lanes
        const float ibr = __shfl(ib, r);                           // item bias of candidate r
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const float ubr = __shfl(ub, 4 * q + reg);             // user bias of user 4 q + reg
            const int64_t urow_i = 16 * tu + 4 * q + reg, ccol = 16 * tc + r;
            if (urow_i < nu && ccol < nc) scores[urow_i * nc + ccol] = acc[reg] + (bias ? ubr + ibr : 0.f);
        }
    }
}

// 64-bit sort keys of the batch: (user slot << 32) | ~rank_key(score): ascending order = users in order, each user's candidates by
// descending score; the stable sort keeps equal scores in candidate order.  Payload = the candidate's position.
__device__ __forceinline__ uint32_t rank_key(float s);
__global__ void batch_keys_kernel(const float* __restrict__ scores, unsigned long long* __restrict__ keys, uint32_t* __restrict__ pos,
                                  int64_t nu, int64_t nc) {
    const int64_t n = nu * nc;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        keys[i] = ((unsigned long long)(i / nc) << 32) | (unsigned long long)(~rank_key(scores[i]));
        pos[i] = (uint32_t)(i % nc);
    }
}

__global__ void take_top_kernel(const uint32_t* __restrict__ spos, const unsigned long long* __restrict__ skeys, int64_t nu, int64_t nc,
                                int64_t topn, int32_t* __restrict__ out_pos, float* __restrict__ out_scores) {
    const int64_t n = nu * topn;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t u = i / topn, k = i % topn;
        out_pos[i] = (int32_t)spos[u * nc + k];
        if (out_scores) {
            const uint32_t key = ~(uint32_t)(skeys[u * nc + k] & 0xFFFFFFFFull);           // rank_key of the score, undone
            out_scores[i] = __builtin_bit_cast(float, (key & 0x80000000u) ? (key ^ 0x80000000u) : ~key);
        }
    }
}

// workspace: [scores nu*nc x 4][keys x 8][sorted keys x 8][positions x 4][sorted positions x 4][histograms of the sort]
int64_t wmf_rank_batch_ws_bytes(int64_t nu, int64_t nc) {
    if (nu <= 0 || nc <= 0) return 256;
    const size_t arr = (((size_t)nu * nc * 4 + 255) / 256) * 256;
    return (int64_t)(7 * arr + wmf_sort_ws_bytes(nu * nc) + 256);
}

int wmf_launch_rank_batch(const float* users, const float* items, int f, int ld, int bias, const int32_t* user_idx, int64_t nu,
                          const int32_t* cand, int64_t nc, int64_t topn, int32_t* out_pos, float* out_scores, void* ws,
                          int64_t ws_bytes, hipStream_t st) {
    (void)f;
    if (nu <= 0 || nc <= 0) return 0;
    if (nu * nc >= (int64_t)1 << 31) return -4;
    if (ws_bytes < wmf_rank_batch_ws_bytes(nu, nc)) return -3;
    const size_t arr = (((size_t)nu * nc * 4 + 255) / 256) * 256;
    char* base = static_cast<char*>(ws);
    float* scores = reinterpret_cast<float*>(base);
    auto* keys = reinterpret_cast<unsigned long long*>(base + arr);
    auto* skeys = reinterpret_cast<unsigned long long*>(base + 3 * arr);
    uint32_t* pos = reinterpret_cast<uint32_t*>(base + 5 * arr);
    uint32_t* spos = reinterpret_cast<uint32_t*>(base + 6 * arr);
    void* temp = base + 7 * arr;
    const int64_t tiles = ((nu + 15) / 16) * ((nc + 15) / 16);
    int64_t grid = (tiles + 3) / 4;
    if (grid > 16384) grid = 16384;
    {
        WmfProfScope ps("score_tile_kernel", st);
        hipLaunchKernelGGL(score_tile_kernel, dim3((unsigned)grid), dim3(256), 0, st, users, items, ld, bias, user_idx, nu, cand, nc,
                           scores);
    }
    hipLaunchKernelGGL(batch_keys_kernel, dim3(2048), dim3(256), 0, st, scores, keys, pos, nu, nc);
    int ubits = 1;
    while ((nu >> ubits) != 0) ++ubits;
    bool in_alt = false;
    if (const int src = wmf_sort_u64(keys, skeys, pos, spos, nu * nc, 32 + ubits, temp, st, &in_alt)) return src;      // (-2 launch failure, -4 too many keys)
    hipLaunchKernelGGL(take_top_kernel, dim3(1024), dim3(256), 0, st, in_alt ? spos : pos, in_alt ? skeys : keys, nu, nc, topn, out_pos,
                       out_scores);
    return 0;
}

__global__ void iota_kernel(int32_t* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = (int32_t)i;
}

int wmf_launch_hits(const float* users, const float* items, int ld, int bias, const int32_t* pair_user,
                    const int32_t* pair_item, const int32_t* pair_row, int64_t n_pairs, const int32_t* cand, int n_cand,
                    const int32_t* slot, const int32_t* topn, int n_topn, int64_t* hits, hipStream_t st) {
    if (hipMemsetAsync(hits, 0, (size_t)n_topn * sizeof(int64_t), st) != hipSuccess) return -2;
    if (n_pairs <= 0) return 0;
    int64_t grid = (n_pairs + 3) / 4;
    if (grid > 16384) grid = 16384;
    WmfProfScope ps("hit_kernel", st);
    hipLaunchKernelGGL(hit_kernel, dim3((unsigned)grid), dim3(256), 0, st, users, items, ld, bias, pair_user, pair_item, pair_row,
                       n_pairs, cand, n_cand, slot, topn, n_topn, reinterpret_cast<unsigned long long*>(hits));
    return 0;
}

// ---- rank of one user: top-n SELECT, as the reference's np.argpartition + argsort of the n best (wmf_model.py:40-43),
// not a sort of the whole candidate list.  Scores become order-preserving 32-bit keys; a 4096-bin histogram of their top
// 12 bits finds the bin T that holds the n-th best; every candidate in a bin >= T -- the n best and the rest of bin T, a
// few thousandths of the list unless the scores pile up -- is compacted as a 64-bit key (score key, ~position) and only
// that short list is sorted (descending: equal scores come out in candidate order, so the result is deterministic although
// the compaction order is not).  One 8-byte read-back tells the host how long the list is: the call synchronises its stream.
__device__ __forceinline__ uint32_t rank_key(float s) {
    const uint32_t u = __builtin_bit_cast(uint32_t, s);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);           // ascending in the float order
}

__global__ __launch_bounds__(256) void rank_hist_kernel(const float* __restrict__ scores, int64_t n, uint32_t* __restrict__ bins) {
    __shared__ uint32_t h[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) h[i] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) atomicAdd(&h[rank_key(scores[i]) >> 20], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < 4096; i += 256) if (h[i]) atomicAdd(&bins[i], h[i]);
}

// ctrl[0] = T: the highest bin such that bins T .. 4095 hold at least topn candidates; ctrl[1] = the compaction counter
__global__ __launch_bounds__(256) void rank_pick_kernel(const uint32_t* __restrict__ bins, int64_t topn, uint32_t* __restrict__ ctrl) {
    __shared__ uint32_t part[256];
    const int t = threadIdx.x;                                   // thread t owns bins 16 t .. 16 t + 15
    uint32_t s = 0;
    for (int i = 0; i < 16; ++i) s += bins[16 * t + i];
    part[t] = s;
    __syncthreads();
    if (t == 0) {
        uint64_t acc = 0;
        int g = 255;
        for (; g > 0 && acc + part[g] < (uint64_t)topn; --g) acc += part[g];
        int b = 16 * g + 15;
        for (; b > 16 * g && acc + bins[b] < (uint64_t)topn; --b) acc += bins[b];
        ctrl[0] = (uint32_t)b;
        ctrl[1] = 0;
    }
}

__global__ __launch_bounds__(256) void rank_compact_kernel(const float* __restrict__ scores, int64_t n, uint32_t* __restrict__ ctrl,
                                                           unsigned long long* __restrict__ keys) {
    const uint32_t T = ctrl[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint32_t k = rank_key(scores[i]);
        // (~k, i): ascending order of these keys = descending score, equal scores in candidate order
        if ((k >> 20) >= T) keys[atomicAdd(&ctrl[1], 1u)] = ((unsigned long long)(~k) << 32) | (unsigned long long)(uint32_t)i;
    }
}

__global__ void rank_take_kernel(const unsigned long long* __restrict__ skeys, const float* __restrict__ scores, int64_t topn,
                                 int32_t* __restrict__ out_pos, float* __restrict__ out_scores) {
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < topn; k += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t pos = (uint32_t)(skeys[k] & 0xFFFFFFFFull);
        out_pos[k] = (int32_t)pos;
        if (out_scores) out_scores[k] = scores[pos];
    }
}

// workspace: [scores n][keys n x 8][sorted keys n x 8][bins 4096 + ctrl][histograms of the sort]
int64_t wmf_rank_ws_bytes(int64_t n) {
    if (n <= 0) return 256;
    const size_t arr = (((size_t)n * 4 + 255) / 256) * 256;
    return (int64_t)(5 * arr + 4096 * 4 + 256 + wmf_sort_ws_bytes(n) + 256);
}

int wmf_launch_rank(const float* users, const float* items, int f, int ld, int bias, const int32_t* user_idx,
                    const int32_t* cand, int64_t n, int64_t topn, int32_t* out_pos, float* out_scores, void* ws,
                    int64_t ws_bytes, hipStream_t st) {
    if (n <= 0) return 0;
    if (ws_bytes < wmf_rank_ws_bytes(n)) return -3;
    const size_t arr = (((size_t)n * 4 + 255) / 256) * 256;
    char* base = static_cast<char*>(ws);
    float* scores = reinterpret_cast<float*>(base);
    auto* keys = reinterpret_cast<unsigned long long*>(base + arr);
    auto* skeys = reinterpret_cast<unsigned long long*>(base + 3 * arr);
    uint32_t* bins = reinterpret_cast<uint32_t*>(base + 5 * arr);
    uint32_t* ctrl = bins + 4096;
    void* temp = base + 5 * arr + 4096 * 4 + 256;
    if (wmf_launch_predict(users, items, f, ld, bias, user_idx, 1, cand, n, scores, st)) return -1;
    if (hipMemsetAsync(bins, 0, 4096 * 4 + 256, st) != hipSuccess) return -2;
    int64_t grid = (n + 255) / 256;
    if (grid > 2048) grid = 2048;
    WMF_LAUNCH("rank_hist_kernel", rank_hist_kernel, dim3((unsigned)grid), dim3(256), 0, st, scores, n, bins);
    WMF_LAUNCH("rank_pick_kernel", rank_pick_kernel, dim3(1), dim3(256), 0, st, bins, topn, ctrl);
    WMF_LAUNCH("rank_compact_kernel", rank_compact_kernel, dim3((unsigned)grid), dim3(256), 0, st, scores, n, ctrl, keys);
    uint32_t host_ctrl[2] = {0, 0};
    if (hipMemcpyAsync(host_ctrl, ctrl, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return -2;
    const size_t m = host_ctrl[1];
    if ((int64_t)m < topn || (int64_t)m > n) return -2;
    bool in_alt = false;
    if (const int src = wmf_sort_u64(keys, skeys, nullptr, nullptr, (int64_t)m, 64, temp, st, &in_alt)) return src;      // (-2 launch failure, -4 too many keys)
    if (!in_alt) skeys = keys;
    hipLaunchKernelGGL(rank_take_kernel, dim3((unsigned)((topn + 255) / 256 > 1024 ? 1024 : (topn + 255) / 256)), dim3(256), 0, st,
                       skeys, scores, topn, out_pos, out_scores);
    return 0;
}
