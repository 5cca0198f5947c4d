// Internal launcher declarations shared by the .hip translation units of libwmf_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define WMF_GRAM_MAX_WAVES 4096
#define WMF_EVAL_MAX_BLOCKS 2048
#define WMF_HEAVY_T 4096      /* rows with more stored entries are accumulated by several waves ... */
#define WMF_SEG 2048          /* ... in segments of this many entries */
#define WMF_WIDE_LU_GRID 64   /* workgroups (and workspace slices) of the pivoted-LU fallback for f > 144 */

extern int wmf_debug_flags;   // kernel-selection switches for timing experiments (tools/kernel_lab.py); 0 in normal use

// row-degree bins of a plan
enum { WMF_BIN_LOW16 = 0, WMF_BIN_LOW32 = 1, WMF_BIN_MFMA = 2, WMF_BIN_GENERAL = 3, WMF_NBINS = 4 };

struct wmf_plan {
    int64_t n;                 // rows
    int f;
    int64_t count[WMF_NBINS];  // rows per bin
    int64_t count8, nnz8;      // rows of the first bin with at most 8 entries (and their entries); they come first in rows[WMF_BIN_LOW16]
    bool bias;                 // created for a biased model: w_eff is allocated (unless split)
    bool split;                // latched at creation: the whitened fixed side comes in the split layout (no w_eff needed)
    mutable int rolled;        // this solve's V / pairs are in the rolled coordinates with the bias bits (wmf_solve_rows_ex; set per call)
    int64_t nnz[WMF_NBINS];    // stored entries per bin
    int32_t* rows[WMF_NBINS];  // device: row ids of each bin (slices of rows_all)
    int32_t* rows_all;         // device: n row ids grouped by bin
    int32_t* fallback_rows;    // device: n slots, rows bounced to the general kernel at run time
    int32_t* fallback_count;   // device: 1 counter
    float* w_eff;              // device: nnz effective weights (values - bias[indices]) of a biased model
    // rows of the MFMA bin with more than WMF_HEAVY_T entries sit at the end of that bin and are split into segments
    int64_t heavy_count, heavy_nnz, seg_total;
    int64_t* seg_lo;           // device: first entry of each segment
    int32_t* seg_d;            // device: entries in each segment
    int32_t* seg_first;        // device: heavy_count + 1 prefix of segment counts
    float* partial;            // device: seg_total x (tiles x 256) partial accumulators
    float* wide_ws;            // device: workspace of the f > 144 pivoted-LU fallback
    // round 4: the first iter_count rows of the heavy bin's ordinary rows have at most wmf_iter_dmax entries: candidates of the
    // matrix-free iteration kernel (wmf_iter.hip); fallback_count[1] counts the rows it hands back in iter_bounce_rows
    int64_t iter_count, iter_nnz;
    int iter_dmax;             // the candidates' longest admissible row at the width / layout the plan was created for
    int32_t* iter_bounce_rows; // device: 2 x iter_count slots (the rows handed back to the elimination kernels | stage 1's hand-on list)
    int32_t* iter_info;        // device: iter_count x {first entry (low, high word), row id, entries}: a candidate's bookkeeping in one 16-byte load
    unsigned long long* iter_stats;   // device: 8 counters (4 reported), accumulated over the launches (wmf_plan_iter_stats reads and clears)
};

int wmf_gram_max_waves(int f);
int wmf_gram_nwaves(int64_t m, int f);
int wmf_launch_gram(const float* Y, int64_t m, int f, int ld, int bias, double* G_sum, float* partial, double* slices,
                    hipStream_t st);
int wmf_launch_factorize(const double* G_sum, int f, int ld, double lambda, float* Wwhite, float* Wunwhite,
                         int32_t* info, double* gA, hipStream_t st);
int wmf_launch_transform(const float* in, int64_t m, int f, int ld, const float* W, int set_col0_one, float* out,
                         float* col0_out, hipStream_t st);

int wmf_launch_solve(const wmf_plan* plan, const float* V, const float* bias_fixed, const int64_t* indptr,
                     const int32_t* indices, const float* values, int f, int ld, float* g, int32_t* fail_count,
                     hipStream_t st);
static inline int wmf_direct_supported(int f) { return f >= 1 && f <= 144; }   // one wave per row holds the f x f system
// Widths whose last feature is a border column of an m-block system (f = 16 m + 1 <= 144, m + 1 not a multiple of 4; debug
// flag 256 switches the border off): k = 16 m with biases.
static inline bool wmf_dw_border(int f) { return f > 16 && f <= 144 && f % 16 == 1 && (f / 16) % 4 != 3 && !(wmf_debug_flags & 256); }
// SPLIT LAYOUT of the whitened fixed side of a bias model at those widths (ld = f + 3): the first f - 1 = 16 m features of
// row i are a packed body row V[i * (f - 1) ..] -- 64 m bytes, whole 128-byte lines when m is even, where an (f + 3)-float
// row at a 528-byte stride (f = 129) touched five lines for 4.03 lines of data -- and the last feature and the side's bias are
// the pair side[2 i], side[2 i + 1] of a small second array (8 bytes per row: 8 MB for a million items, resident in L2 /
// Infinity Cache).  The row kernels subtract the bias from the entry weights as it arrives (RecModel/wmf_model.py:343): no
// separate pass over all entries.  Written by wmf_row_transform(set_col0_one = 1), read by every row kernel of
// wmf_solve_rows / wmf_accumulate_rows when a bias array is given.
static inline bool wmf_split_layout(int f, int ld) { return wmf_dw_border(f) && ld == f + 3; }
// wmf_directl.hip: normal heavy rows at f = 128 / 129 through an LDS-DMA row ring
int wmf_directl_supported(int f, int ld);
// (side: NULL, or the {last feature, bias} pairs of the split layout; V is then the packed body)
// (count_dev: NULL, or the device-side number of rows; count is then the capacity of the list -- the bounce list of wmf_iter.hip)
int wmf_launch_directl(const int32_t* rows, int64_t count, const float* V, const float* side, const int64_t* indptr,
                       const int32_t* indices, const float* vals, int f, int ld, float* g, int32_t* fb_rows, int32_t* fb_count,
                       hipStream_t st, const int32_t* count_dev = nullptr);
int wmf_launch_directl_segments(int64_t nseg, const float* V, const float* side, const int32_t* indices, const float* vals, int f,
                                int ld, const int64_t* seg_lo, const int32_t* seg_d, float* partial, hipStream_t st);
int wmf_launch_directw(const wmf_plan* pl, const float* V, const float* side, const int64_t* indptr,
                       const int32_t* indices, const float* vals, int f, int ld, float* g, hipStream_t st);
int64_t wmf_directw_partial_floats(int f);
int wmf_launch_accumulate(const float* V, const float* side, const int64_t* indptr, const int32_t* degrees, const int32_t* indices,
                          const float* vals, int64_t n, int f, int ld, float* partial, int slot_stride, int slot_offset,
                          hipStream_t st);
int wmf_launch_eliminate(float* partial, int64_t n, int slots_per_row, int f, int ld, float* g, int32_t* fb_rows,
                         int32_t* fail_count, hipStream_t st);
void wmf_launch_bias_adjust(const float* vals, const int32_t* indices, const float* biasv, int64_t nnz, float* w_eff,
                            hipStream_t st);
// wmf_csr.hip: COO -> CSR, stable in (row, column)
int64_t wmf_csr_ws_bytes(int64_t nnz, int64_t n_rows, int64_t n_cols);
int wmf_launch_coo_to_csr(const int64_t* rows, const int64_t* cols, const float* vals, int64_t nnz, int64_t n_rows, int64_t n_cols,
                          int64_t* indptr, int32_t* indices, float* values, int32_t* bad_flag, void* ws, int64_t ws_bytes,
                          hipStream_t st);
int wmf_wide_supported(int f);
int wmf_launch_wide(const int32_t* rows, int64_t count, const float* V, const float* biasv, const int64_t* indptr,
                    const int32_t* indices, const float* vals, int f, int ld, float* g, int32_t* fb_rows,
                    int32_t* fb_count, hipStream_t st, const int32_t* count_dev = nullptr);
// wmf_iter.hip: rows with 33 .. wmf_iter_dmax entries whose whitened system is close to the identity, by a matrix-free
// Neumann / Chebyshev iteration; the rows it does not solve are appended to bounce_rows (count on the device)
int wmf_iter_dmax(int f, int ld, int split);
int wmf_launch_iter(const int32_t* rows, int64_t count, const float* V, const float* side, const int64_t* indptr,
                    const int32_t* indices, const float* vals, int f, int ld, float* g, int32_t* bounce_rows,
                    int32_t* bounce_count, unsigned long long* stats, const void* info, hipStream_t st, int lsb = 0);
static inline bool wmf_iter_enabled() { return !(wmf_debug_flags & 268435456); }
// The rolled whitened coordinates with the bias in the body's last mantissa bits (include/wmf_hip.h, wmf_row_transform modes 3 / 4,
// wmf_solve_rows_ex): bias models whose packed body is 128 floats (k = 128), transform6_kernel and the iteration kernels only
static inline bool wmf_rolled_layout(int f, int ld) {
    return f == 129 && wmf_split_layout(f, ld) && !(wmf_debug_flags & 262144) && !(wmf_debug_flags & 1073741824);
}
// candidates of this call: none when the iteration is switched off, or when the call's layout (ld, split) is not the one the
// plan sorted its rows for (a caller with its own leading dimension: the kernel's register slots would not hold the rows)
static inline int64_t wmf_iter_rows(const wmf_plan* pl, int f, int ld, bool split) {
    return (wmf_iter_enabled() && pl->iter_count > 0 && wmf_iter_dmax(f, ld, split ? 1 : 0) >= pl->iter_dmax) ? pl->iter_count : 0;
}
int wmf_rowsplit_supported(int f);
int wmf_launch_rowsplit(const wmf_plan* pl, const float* V, const float* biasv, const int64_t* indptr,
                        const int32_t* indices, const float* vals, int f, int ld, float* g, hipStream_t st);
int64_t wmf_rowsplit_partial_floats(int f);      // floats of one segment's partial system
// partial systems of the segments of every split row summed, in segment order, into the row's first slot (wmf_solve.hip)
void wmf_launch_combine_segments(const wmf_plan* pl, int64_t partial_floats, hipStream_t st);
size_t wmf_wide_lu_workspace_bytes(int f);
int wmf_launch_wide_lu(const int32_t* rows, const int32_t* count_ptr, const float* V, const float* biasv,
                       const int64_t* indptr, const int32_t* indices, const float* vals, int f, int ld, float* g,
                       int32_t* fail_count, float* work, hipStream_t st);
int wmf_launch_spmm(const float* V, const int64_t* indptr, const int32_t* indices, const float* values, int64_t n,
                    int ld, float* g, hipStream_t st);
int wmf_launch_eval(const float* users, const float* items, int f, int ld, int bias, const int64_t* indptr,
                    const int32_t* indices, const float* values, int64_t n, double* out3, double* partial,
                    hipStream_t st);
int wmf_launch_predict(const float* users, const float* items, int f, int ld, int bias, const int32_t* ui, int64_t n_u,
                       const int32_t* ii, int64_t n_i, float* out, hipStream_t st);
int wmf_launch_hits(const float* users, const float* items, int ld, int bias, const int32_t* pair_user,
                    const int32_t* pair_item, const int32_t* pair_row, int64_t n_pairs, const int32_t* cand, int n_cand,
                    const int32_t* slot, const int32_t* topn, int n_topn, int64_t* hits, hipStream_t st);
int64_t wmf_rank_ws_bytes(int64_t n);
int wmf_launch_rank(const float* users, const float* items, int f, int ld, int bias, const int32_t* user_idx,
                    const int32_t* cand, int64_t n, int64_t topn, int32_t* out_pos, float* out_scores, void* ws,
                    int64_t ws_bytes, hipStream_t st);
int64_t wmf_rank_batch_ws_bytes(int64_t nu, int64_t nc);
int wmf_launch_rank_batch(const float* users, const float* items, int f, int ld, int bias, const int32_t* user_idx, int64_t nu,
                          const int32_t* cand, int64_t nc, int64_t topn, int32_t* out_pos, float* out_scores, void* ws,
                          int64_t ws_bytes, hipStream_t st);
int wmf_launch_gather_rows(const float* in, int ld, const int64_t* rows, int64_t n, float* out, hipStream_t st);
int wmf_launch_confidence(float* values, int64_t nnz, double alpha, double beta, int mode, hipStream_t st);
// float64 half step of the cores > 1 variants (wmf_f64.hip)
int64_t wmf_f64_ws_bytes(int f, int64_t m, int64_t n);
int wmf_launch_half_step_f64(const double* Y, int64_t m, int f, int bias, const int64_t* indptr, const int32_t* indices,
                             const double* values, int64_t n, double lambda, double* X, void* ws, int32_t* fail, hipStream_t st);
int wmf_launch_confidence_f64(double* values, int64_t nnz, double alpha, double beta, int mode, hipStream_t st);
// wmf_iter64.hip: the matrix-free iteration in float64 (rows of 1 .. 32 entries: low = 1; 33 .. wmf_iter64_dmax: low = 0)
int wmf_iter64_dmax(int f);
int wmf_launch_iter64(const double* V, const double* Y, int f, int bias, const int64_t* indptr, const int32_t* indices,
                      const double* vals, int64_t n, int low, double* gout, int32_t* state, const int32_t* ctrl, hipStream_t st);

void wmf_set_error(const char* fmt, ...);
// Ablation switches whose results are WRONG (1 no elimination, 2 no accumulation MFMAs, 8 no tile inverse) are compiled
// into a -DWMF_LAB build only (tools/build_variant.sh); the shipped library has no such code path.
#ifdef WMF_LAB
#define WMF_ABL(dbg, bits) ((dbg) & (bits))
#else
#define WMF_ABL(dbg, bits) 0
#endif

// per-kernel event timing (wmf_api.hip).  A launch site names its kernel the way rocprofv3 prints the symbol, up to and
// including the template arguments ("solve_low_kernel<9, 1, false, false>"), so that bench.py's table and a
// rocprofv3 --kernel-trace --stats summary of the same run can be matched line by line.
const char* wmf_kname(const char* fmt, ...);     // interned: the pointer stays valid for the life of the library
void wmf_prof_begin(const char* name, hipStream_t st);
void wmf_prof_end(hipStream_t st);
struct WmfProfScope {
    hipStream_t st;
    WmfProfScope(const char* name, hipStream_t s) : st(s) { wmf_prof_begin(name, s); }
    ~WmfProfScope() { wmf_prof_end(st); }
};
#define WMF_LAUNCH(NAME, KERNEL, GRID, BLOCK, LDS, ST, ...)                          \
    do {                                                                             \
        WmfProfScope wmf_ps_(NAME, ST);                                              \
        hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, ST, __VA_ARGS__);               \
    } while (0)
