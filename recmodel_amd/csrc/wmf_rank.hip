// Ranking-side kernels (gfx950), SURVEY.md section 8(f) row 4:
//   * hit_kernel   -- RecModel.eval_topn / compute_hit (RecModel/base_model.py:51-148): for every test entry
//                     (user, item) the reference ranks the item among `rand_sampled` random candidates with
//                     WMF.rank and asks whether it made the top n.  The item makes the top n exactly when fewer
//                     than n candidates score higher, so one wave per test entry scores the candidates (16 lanes
//                     per candidate, four at a time) and counts; hits are integer atomics, hence deterministic.
//                     The random draws stay on the host, in the reference's np.random call order.
//   * rank         -- WMF.rank (RecModel/wmf_model.py:25-47): scores of one user against a candidate list, then a
//                     stable descending radix sort of (score, position) pairs (rocPRIM's device sort: the ordering
//                     is not the hot path; the scores are the same predict kernel as eval_prec).
#include <cstring>            // rocprim/iterator/texture_cache_iterator.hpp uses memset without including it

#include <rocprim/rocprim.hpp>

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
    WmfProfScope ps(WMF_SLOT_PREDICT, st);
    hipLaunchKernelGGL(hit_kernel, dim3((unsigned)grid), dim3(256), 0, st, users, items, ld, bias, pair_user, pair_item, pair_row,
                       n_pairs, cand, n_cand, slot, topn, n_topn, reinterpret_cast<unsigned long long*>(hits));
    return 0;
}

// workspace: [scores n][sorted scores n][positions n][sorted positions n][rocPRIM temporary]
static size_t rank_sort_temp_bytes(int64_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs_desc(nullptr, bytes, (float*)nullptr, (float*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr,
                                         (size_t)n, 0, 32, (hipStream_t)0);
    return bytes;
}

int64_t wmf_rank_ws_bytes(int64_t n) {
    if (n <= 0) return 256;
    const size_t arr = (((size_t)n * 4 + 255) / 256) * 256;
    return (int64_t)(4 * arr + rank_sort_temp_bytes(n) + 256);
}

int wmf_launch_rank(const float* users, const float* items, int f, int ld, int bias, const int32_t* user_idx,
                    const int32_t* cand, int64_t n, int64_t topn, int32_t* out_pos, float* out_scores, void* ws,
                    int64_t ws_bytes, hipStream_t st) {
    if (n <= 0) return 0;
    if (ws_bytes < wmf_rank_ws_bytes(n)) return -3;
    const size_t arr = (((size_t)n * 4 + 255) / 256) * 256;
    char* base = static_cast<char*>(ws);
    float* scores = reinterpret_cast<float*>(base);
    float* sorted = reinterpret_cast<float*>(base + arr);
    int32_t* pos = reinterpret_cast<int32_t*>(base + 2 * arr);
    int32_t* spos = reinterpret_cast<int32_t*>(base + 3 * arr);
    void* temp = base + 4 * arr;
    size_t temp_bytes = rank_sort_temp_bytes(n);
    if (wmf_launch_predict(users, items, f, ld, bias, user_idx, 1, cand, n, scores, st)) return -1;
    int64_t grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(iota_kernel, dim3((unsigned)grid), dim3(256), 0, st, pos, n);
    if (rocprim::radix_sort_pairs_desc(temp, temp_bytes, scores, sorted, pos, spos, (size_t)n, 0, 32, st) != hipSuccess) return -2;
    if (hipMemcpyAsync(out_pos, spos, (size_t)topn * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return -2;
    if (out_scores && hipMemcpyAsync(out_scores, sorted, (size_t)topn * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return -2;
    return 0;
}
