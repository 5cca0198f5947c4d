/* wmf_hip.h -- C ABI of libwmf_hip.so: MI355X (gfx950) kernels for the WMF/ALS hot path.
 *
 * The reference (titoeb/RecModel) has no FFI or operator interface for this path: the seam is
 * the pair of pure functions
 *
 *     WMF.recompute_factors(Y, C, lambda_reg)            RecModel/wmf_model.py:213-240
 *     WMF.recompute_factors_bias(Y, C, lambda_reg, ...)  RecModel/wmf_model.py:311-351
 *
 * plus WMF.predict (wmf_model.py:191-211) as used by RecModel.eval_prec (base_model.py:150-179).
 * This header declares what a ctypes / cffi binding for those functions binds.  INTEGRATION.md
 * shows the reference-side stub.
 *
 * Conventions
 *   - plain C types only; every function returns 0 on success or a negative WMF_E* code, and
 *     wmf_last_error() returns a message for the calling thread's last failure.
 *   - "host" entry points take host pointers and are self-contained (own hipMalloc/copies).
 *   - "device" entry points take device pointers and a hipStream_t passed as void*; they only
 *     enqueue work (no allocation, no synchronisation) so a caller can keep factors resident in
 *     HBM across half steps and put collectives between them.  Whatever device memory a solve needs
 *     beyond the caller's buffers is allocated by wmf_plan_create and freed by wmf_plan_destroy.
 *   - factor matrices are row-major fp32 with a leading dimension `ld` (floats), ld % 4 == 0,
 *     ld >= f; the padding columns [f, ld) must be zero on input and are written as zero.
 *     f = k (no bias) or k+1 (bias: column 0 is the bias, wmf_model.py:328-331).
 *   - CSR is consumed "as stored": stored zeros contribute, duplicates are not merged
 *     (wmf_model.py:231-239).  indptr is int64, indices int32, values fp32.
 */
#ifndef WMF_HIP_H
#define WMF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WMF_OK            0
#define WMF_EINVAL       -1   /* bad argument (shape, alignment, unsupported f)           */
#define WMF_EHIP         -2   /* a HIP runtime call failed                                */
#define WMF_ENOMEM       -3   /* allocation failed                                        */
#define WMF_ENUMERIC     -4   /* Gramian not positive definite / singular row system      */

#define WMF_MAX_F        260  /* largest supported factor width (k = 256 + bias, padded)  */

typedef struct wmf_plan wmf_plan;   /* opaque: degree-binned row schedule of one CSR matrix */

/* ---- library ---------------------------------------------------------------------------- */
const char* wmf_last_error(void);
int  wmf_version(void);
/* Smallest legal leading dimension for factor width f (round up to a multiple of 4). */
int  wmf_ld_for(int f);

/* ---- host-level drop-in for the operator seam -------------------------------------------- */
/* X_new = recompute_factors[_bias](Y, C, lambda)          wmf_model.py:213-240 / 311-351
 *   Y_host   [m, f]  row-major contiguous fp32 (bias != 0: column 0 is the fixed side's bias)
 *   C        CSR [n, m]: indptr int64[n+1], indices int32[nnz], values fp32[nnz]
 *   X_host   [n, f]  output, row-major contiguous fp32
 * Runs on the current HIP device; synchronous. */
int wmf_recompute_factors_host(const float* Y_host, int64_t m, int f, int bias,
                               const int64_t* indptr, const int32_t* indices, const float* values,
                               int64_t n, double lambda, float* X_host);

/* ---- device-level building blocks (resident training loop) ------------------------------- */
/* Workspace (device, caller-allocated) needed by the Gramian + factorisation step. */
int64_t wmf_gram_workspace_bytes(int f);

/* Partial Gramian of a (shard of a) factor matrix:  G_sum[f*f] (fp64, row-major) =
 * sum_r y~_r y~_r^T over the m rows given, where y~ = y with column 0 replaced by 1 when
 * bias != 0 (wmf_model.py:331-332).  No lambda yet: multi-GPU callers all-reduce G_sum first.
 * wmf_model.py:215 (np.dot(Y.T, Y)). */
int wmf_gram(const float* Y, int64_t m, int f, int ld, int bias,
             double* G_sum, void* workspace, void* stream);

/* From the (all-reduced) Gramian: A = G_sum + lambda*I = L L^T (fp64, on device) and the two
 * fp32 transforms used by the row solve, both [f, ld] row-major, zero padded:
 *   W_white[a][b]   = (L^-T)[a][b]     V = Y~ . W_white      (whitened fixed factors)
 *   W_unwhite[a][b] = (L^-1)[a][b]     X = g  . W_unwhite    (back to factor space)
 * info (device int32, caller zeroes): left alone on success, set to j > 0 when leading minor j is not positive
 * definite -- sticky, so one check after several half steps still sees a failure of the first (the outputs of a
 * failed factorisation are zero matrices).
 * Replaces the `+ lambda_reg * np.eye(...)` of wmf_model.py:215/332 and the shared part of the
 * per-row np.linalg.solve of :239/:350. */
int wmf_factorize(const double* G_sum, int f, int ld, double lambda,
                  float* W_white, float* W_unwhite, int32_t* info,
                  void* workspace, void* stream);

/* out[r, :] = in~[r, :] . W  for rows [0, m), in/out [m, ld], W [f, ld]; padding columns are written as zero.
 * set_col0_one != 0 (whitening of the fixed side of a bias model): in~ is `in` with column 0 read as 1, and the bias
 * in[r,0] (wmf_model.py:328-331) is copied to col0_out[r] (may be NULL).
 * SPLIT LAYOUT.  For the widths with wmf_whitened_row_floats(f, ld, 1) == f - 1 -- f = 16 m + 1 <= 144 with ld = f + 3:
 * k = 16, 32, 64, 80, 96, 128 with biases -- set_col0_one != 0 writes the whitened side in the form the row kernels gather
 * cheapest: out is a PACKED body, out[r * (f - 1) + c] = feature c < f - 1 of row r (64 m bytes a row, whole 128-byte lines,
 * where the (f + 3)-float row at its 528-byte stride touched five lines for 4.03 lines of data), and col0_out, then
 * REQUIRED, is float[2 m]: col0_out[2 r] = feature f - 1, col0_out[2 r + 1] = the bias.  out may not alias in there.
 * wmf_solve_rows / wmf_accumulate_rows take the two arrays as their V and bias_fixed. */
/* ROLLED COORDINATES (round 4; wmf_rolled_layout_supported(f, ld) != 0: bias models with a 128-float body, k = 128).  With a
 * bias model whitened feature 0 is the same number for every row -- in~'s column of ones times an upper-triangular W_white -- so
 * a caller may keep the whitened side in coordinates rolled by one: feature c at position c - 1, feature 0 last.
 *   set_col0_one = 3  whitening as above (split layout), written in rolled coordinates: the border feature of the pairs is
 *                     then that constant, and the 32 bits of the row's bias replace the last mantissa bit of body positions
 *                     8 j, 8 j + 1 (j < 16; each value moves by at most one ulp).  The row kernels that hold eight consecutive
 *                     features per lane rebuild the bias from the row they gathered: the pairs -- 8 bytes that cost a whole
 *                     128-byte line per stored entry, 1.0 - 1.9 ms of configs[2]'s item half step -- are not fetched at all.
 *   set_col0_one = 4  the input is in rolled coordinates (the g of wmf_solve_rows_ex(WMF_SOLVE_ROLLED)): out = in . W as for 0.
 * Every kernel accepts such a V / pairs (it is a permutation of the features); wmf_solve_rows_ex must be told, because the
 * kernels that skip the pairs rely on both properties. */
int wmf_rolled_layout_supported(int f, int ld);

int wmf_row_transform(const float* in, int64_t m, int f, int ld, const float* W,
                      int set_col0_one, float* out, float* col0_out, void* stream);
/* Floats per row of the whitened fixed side that wmf_row_transform(set_col0_one = bias) writes and the row kernels read:
 * ld, or f - 1 for the split layout above (0: f, ld not supported). */
int wmf_whitened_row_floats(int f, int ld, int bias);

/* Degree-binned schedule for the rows of one CSR matrix (built once per matrix, host indptr).  Allocates (and
 * synchronously fills) the plan's device memory, including everything wmf_solve_rows will need: the row lists, the
 * segment table of rows with more than 4096 entries, with bias != 0 the bias-adjusted weights (one float per stored
 * entry, wmf_model.py:343), for f > 144 the workspace of the pivoted fallback.  A plan created with bias = 0 cannot
 * be solved with a bias vector. */
int  wmf_plan_create(const int64_t* indptr_host, int64_t n, int f, int bias, wmf_plan** out);
void wmf_plan_destroy(wmf_plan* p);
/* Rows and stored entries the plan routes to each kernel family: out12[b] = rows, out12[4+b] =
 * stored entries, for b = 0: <=16 entries, 1: 17..32, 2: one wave per row (f <= 144), 3: four waves per row (f > 144);
 * out12[8], out12[9] = rows and entries of bin 0 with at most 8 entries (two rows share a wave);
 * out[10], out[11] = rows and entries of bin 2 / 3 with more than 4096 entries (split over several waves);
 * out[12], out[13] = rows and entries of bin 2 / 3 that are candidates of the matrix-free iteration kernel (33 .. a
 * width-dependent number of entries; round 4).  The array has 14 elements. */
int  wmf_plan_stats(const wmf_plan* p, int64_t* out14);
/* What became of those candidates in the solves since the last call: out4 = { rows solved by the iteration, rows handed
 * back to the elimination kernels (bound on the row's operator too weak, or no convergence), applications of the row
 * operator in total, rows solved by the Chebyshev rather than the Neumann recurrence }.  Data dependent: a row is solved
 * by a truncated polynomial in E = V_u^T D V_u only when tr E bounds it away from the direct method's cost
 * (csrc/wmf_iter.hip).  Synchronises the device and clears the counters. */
int  wmf_plan_iter_stats(wmf_plan* p, int64_t* out4);

/* The per-row normal-equation solve in whitened coordinates, for every row of the CSR:
 *   g_u = (I + V_u^T D_u V_u)^-1 V_u^T (w_u + 1),   V_u = V[idx_u], D_u = diag(w_u)
 * with w_u = values - bias_fixed[idx_u] when bias_fixed != NULL (wmf_model.py:343).  V and bias_fixed must both come
 * from wmf_row_transform(set_col0_one = 1) on the same matrix (for the widths named there: packed body and pairs; the
 * kernels then take the bias with the gathered row, elsewhere one pass over the entries folds it into the weights).
 * g [n, ld]; follow with wmf_row_transform(g, W_unwhite) to obtain X_new.
 * Restates the loop body wmf_model.py:220-239 / 337-350.  Rows without stored entries give 0.
 * fail_count (device int32, caller zeroes): number of rows whose system was numerically singular. */
int wmf_solve_rows(const wmf_plan* plan, const float* V, const float* bias_fixed,
                   const int64_t* indptr, const int32_t* indices, const float* values,
                   int64_t n, int f, int ld, float* g, int32_t* fail_count, void* stream);
/* The same, with flags: WMF_SOLVE_ROLLED = V and bias_fixed come from wmf_row_transform(set_col0_one = 3) (see there); g is
 * then in rolled coordinates: follow with wmf_row_transform(g, W_unwhite, set_col0_one = 4). */
#define WMF_SOLVE_ROLLED 1
int wmf_solve_rows_ex(const wmf_plan* plan, const float* V, const float* bias_fixed,
                   const int64_t* indptr, const int32_t* indices, const float* values,
                   int64_t n, int f, int ld, float* g, int32_t* fail_count, int flags, void* stream);

/* Sum over the stored, non-zero entries of a CSR "utility" matrix of (value - predict(u, i))^2
 * and |value - predict(u, i)| -- RecModel.eval_prec (base_model.py:163-176) with WMF.predict
 * (wmf_model.py:205-211).  The CSR has n user rows; `users` is the matching [n, ld] factor block
 * (a multi-GPU caller passes its own user shard and the shard's CSR), `items` the full item
 * matrix.  Stored zeros are skipped (utility_mat.nonzero(), base_model.py:163).
 * out3 (device fp64[3]) = { sum of squares, sum of abs, count }.
 * workspace: device, wmf_eval_workspace_bytes() bytes. */
int64_t wmf_eval_workspace_bytes(void);
int wmf_eval_sqerr(const float* users, const float* items, int f, int ld, int bias,
                   const int64_t* indptr, const int32_t* indices, const float* values,
                   int64_t n, double* out3, void* workspace, void* stream);

/* out[p] = predict(users_idx[p], items_idx[p])   wmf_model.py:205-211.
 * n_u == 1 or n_i == 1 broadcasts that index (the reference's one-user / one-item form). */
int wmf_predict_pairs(const float* users, const float* items, int f, int ld, int bias,
                      const int32_t* users_idx, int64_t n_u, const int32_t* items_idx, int64_t n_i,
                      float* out, void* stream);

/* WMF.rank, wmf_model.py:25-47: scores of ONE user (*user_idx, device) against n_cand candidate item rows
 * (cand_idx, device), the topn best sorted by score, best first.  out_pos[k] = position in cand_idx of the k-th best
 * candidate (the reference returns items[order]); ties keep candidate order (the reference's tie order is
 * whatever argpartition / argsort leave).  out_scores may be NULL.  1 <= topn <= n_cand.
 * A top-n select like the reference's argpartition (:40-43), not a sort of all candidates: only the candidates of the
 * score-histogram bins that reach down to the topn-th best are sorted.  The length of that list is read back, so this
 * call SYNCHRONISES its stream.  workspace: wmf_rank_workspace_bytes(n_cand) bytes of device memory. */
int64_t wmf_rank_workspace_bytes(int64_t n_cand);
int wmf_rank_topn(const float* users, const float* items, int f, int ld, int bias, const int32_t* user_idx,
                  const int32_t* cand_idx, int64_t n_cand, int64_t topn, int32_t* out_pos, float* out_scores,
                  void* workspace, int64_t workspace_bytes, void* stream);

/* WMF.rank with a LIST of users (wmf_model.py:29-32 loops; here one launch): every user in user_idx (n_users,
 * device) against the same candidate list; out_pos[u * topn + k] = position in cand_idx of user u's k-th best.
 * Scores by f32 MFMA (16 users x 16 candidates per wave), ordering by a stable segmented radix sort.
 * n_users * n_cand < 2^31; workspace: wmf_rank_batch_workspace_bytes(n_users, n_cand). */
int64_t wmf_rank_batch_workspace_bytes(int64_t n_users, int64_t n_cand);
int wmf_rank_topn_batch(const float* users, const float* items, int f, int ld, int bias, const int32_t* user_idx,
                        int64_t n_users, const int32_t* cand_idx, int64_t n_cand, int64_t topn, int32_t* out_pos,
                        float* out_scores, void* workspace, int64_t workspace_bytes, void* stream);

/* RecModel.eval_topn / compute_hit, base_model.py:51-148, for all test entries at once.
 * Test entry p = (pair_user[p], pair_item[p]); its user's random candidates are row pair_row[p] of
 * candidates[n_rows][n_cand] (item rows, drawn by the caller exactly as base_model.py:62-63 draws them) and
 * position slot[pair_row[p]] of that row is where the reference writes the test item (:84).
 * hits[t] = number of test entries whose item is among the topn[t] best of its candidate row, i.e. for
 * which fewer than topn[t] of the other n_cand - 1 candidates score strictly higher (:86-93).
 * All arrays on the device; hits is int64[n_topn], n_topn <= 64. */
int wmf_hit_counts(const float* users, const float* items, int f, int ld, int bias,
                   const int32_t* pair_user, const int32_t* pair_item, const int32_t* pair_row, int64_t n_pairs,
                   const int32_t* candidates, int32_t n_cand, const int32_t* slot,
                   const int32_t* topn, int32_t n_topn, int64_t* hits, void* stream);

/* out[i, :] = in[rows[i], :] for i < n: the rows of a freshly solved factor block packed per destination rank, for the
 * need-list exchange of the sharded engine (the parallelism that replaces the reference's Pool, wmf_model.py:245-249): each
 * rank is sent only the rows its own half step reads.  in / out row-major with ld floats per row (ld a multiple of 4),
 * rows int64[n] (indices into `in`; validated by the caller), all on the device.  Enqueues only. */
int wmf_gather_rows(const float* in, int ld, const int64_t* rows, int64_t n, float* out, void* stream);

/* g[u, :] = sum_j values[j] * V[indices[j], :]  (CSR x dense), the SpMM of the un-weighted
 * closed form wmf_model.py:85,88 in whitened coordinates. */
int wmf_spmm_rows(const float* V, const int64_t* indptr, const int32_t* indices, const float* values,
                  int64_t n, int f, int ld, float* g, void* stream);

/* COO -> CSR on the device, stable in (row, column): entry e = (rows[e], cols[e], values[e]), all device arrays.  With
 * (rows, cols) swapped this is the transpose the reference takes before training (wmf_model.py:128, count_mat.T.tocsr());
 * the sharded engine also builds every rank's shards with it.  Duplicates are kept, in their stored order.
 * indptr_out int64[n_rows + 1], indices_out int32[nnz], values_out float[nnz]; n_cols <= 2^31 - 1, n_rows * n_cols < 2^63,
 * nnz < 2^32.  bad_flag (device int32, caller zeroes): set to 1 if an entry lies outside [0, n_rows) x [0, n_cols) -- the
 * outputs are then unspecified (never written out of bounds).  workspace: wmf_coo_to_csr_workspace_bytes() bytes. */
int64_t wmf_coo_to_csr_workspace_bytes(int64_t nnz, int64_t n_rows, int64_t n_cols);
int wmf_coo_to_csr(const int64_t* rows, const int64_t* cols, const float* values, int64_t nnz, int64_t n_rows, int64_t n_cols,
                   int64_t* indptr_out, int32_t* indices_out, float* values_out, int32_t* bad_flag, void* workspace,
                   int64_t workspace_bytes, void* stream);

/* values[i] = alpha*log(1+beta*values[i]) (mode 0) or alpha*values[i] (mode 1), in place on the
 * device.  wmf_model.py:119-123. */
int wmf_confidence_transform(float* values, int64_t nnz, double alpha, double beta, int mode, void* stream);
/* The same on float64 values (the count matrix of a `cores > 1` run stays float64, wmf_model.py:119-123). */
int wmf_confidence_transform_f64(double* values, int64_t nnz, double alpha, double beta, int mode, void* stream);

/* ---- float64 half step: the reference's Pool variants (a5 / a6) ------------------------------------------------------
 * recompute_factors_par / recompute_factors_bias_par (wmf_model.py:242-265) and their row functions
 * recompute_factors_intern / recompute_factors_bias_intern (:267-309): with a float64 count matrix every row is
 *   x_u = solve(Y~^T Y~ + lambda I + Y~_u^T diag(w) Y~_u,  Y~_u^T (w + 1))        in float64,
 * Y~ = Y with column 0 read as 1 and w = c_u - Y[idx, 0] for bias != 0 (:253-257, :279), rows without stored entries
 * are zero (:274-276, :296-298), and the float64 rows are stacked without a cast (np.stack, :246 / :261) -- so the
 * reference's `cores > 1` training continues on float64 factors (cores = 4 is the reference's default, :49-51).  This entry
 * point is that arithmetic on the device, without the whitening of the float32 path: Gramian in float64, then per row
 * the system accumulated and factored in registers, one workgroup or wave per row (blocked Cholesky; rows with at most 32 stored
 * entries through the whitened low-rank form -- a d x d system -- when they are at least a quarter of the rows; a system that is not positive
 * definite -- bias-adjusted weights below zero -- or not finite is redone by LU with partial pivoting, np.linalg.solve's
 * gesv); both agree with the float64 reference arithmetic to 1e-10.
 *   Y [m, f] float64 row-major (dense, no padding), values float64[nnz], X [n, f] float64 out, all on the device;
 *   workspace: wmf_half_step_f64_workspace_bytes(f, m, n) bytes; fail_count (device int32, caller zeroes): += 1 per
 *   exactly singular row system (its X row is NaN; the reference raises LinAlgError).  Enqueues only. */
int64_t wmf_half_step_f64_workspace_bytes(int f, int64_t m, int64_t n);
int wmf_half_step_f64(const double* Y, int64_t m, int f, int bias, const int64_t* indptr, const int32_t* indices,
                      const double* values, int64_t n, double lambda, double* X, void* workspace, int64_t workspace_bytes,
                      int32_t* fail_count, void* stream);
/* The same with host buffers in and out (what recompute_factors_par(Y, C, lambda_reg, cores) is handed, :242):
 * synchronous, allocates and frees its device buffers; WMF_ENUMERIC if a row system was singular. */
int wmf_recompute_factors_f64_host(const double* Y_host, int64_t m, int f, int bias, const int64_t* indptr,
                                   const int32_t* indices, const double* values, int64_t n, double lambda, double* X_host);

/* ---- partial systems for a reduce-scatter exchange (f <= 144) ------------------------------------
 * When the rows being updated are few and the fixed side is large (many users, few items), gathering the fixed
 * side's factors costs more than exchanging the rows' accumulated systems: every rank accumulates, for EVERY row,
 * the part of  V_u^T D V_u  and  V_u^T p  that its own slice of the fixed side contributes, the partial systems are
 * summed over ranks (reduce-scatter), and each rank eliminates the rows it owns.  Same arithmetic as
 * wmf_solve_rows' heavy-row kernel (wmf_model.py:233-239), split at the sum over stored entries.
 *   wmf_partial_row_floats(f)  floats per row of a partial-system buffer (0: width not supported);
 *   wmf_accumulate_rows        slot (i * slot_stride + slot_offset) of `partial` = system of CSR row i over the
 *                              entries given (zeros for an empty row).  slot_stride = 1, slot_offset = 0: one slot per
 *                              row.  Larger strides let several calls -- one per arriving chunk of the fixed side --
 *                              fill different slots of the same rows.  degrees = int32[n] device,
 *                              indptr[i + 1] - indptr[i]; bias_fixed != NULL needs w_eff_workspace (float[nnz], device)
 *                              unless V / bias_fixed are in the split layout of wmf_row_transform;
 *   wmf_eliminate_rows         g[i] = (I + sum of row i's slots_per_row consecutive slots)^-1 (sum of their y), slots
 *                              added in order; a row whose system is not positive definite is counted in fail_count
 *                              (there is no CSR here to hand to the pivoted fallback); scratch = int32[n] device. */
int64_t wmf_partial_row_floats(int f);
int wmf_accumulate_rows(const float* V, const float* bias_fixed, const int64_t* indptr, const int32_t* degrees,
                        const int32_t* indices, const float* values, int64_t n, int64_t nnz, int f, int ld,
                        float* partial, int32_t slot_stride, int32_t slot_offset, float* w_eff_workspace, void* stream);
int wmf_eliminate_rows(float* partial, int64_t n, int32_t slots_per_row, int f, int ld, float* g, int32_t* fail_count,
                       int32_t* scratch, void* stream);

/* ---- per-kernel timing (bench.py roofline) ------------------------------------------------- */
/* While enabled, every kernel launch of the device-level entry points is bracketed by two HIP events on its own
 * stream.  wmf_profile_collect() waits for the events recorded so far, folds them into a table with one entry per
 * (kernel symbol, tag) and returns the number of entries; wmf_profile_entry(i, ...) reads entry i: the kernel's name as
 * rocprofv3 prints it, up to and including its template arguments ("solve_low_kernel<9, 1, false, false>"), the tag,
 * total / min / max milliseconds and the launch count.  The tag is whatever wmf_profile_set_tag() last set (bench.py:
 * 0 = users half step, 1 = items half step), so the launches of one kernel on the two sides are kept apart.
 * wmf_profile_reset() empties the table.  All five are thread-safe. */
int wmf_profile_enable(int on);
int wmf_profile_set_tag(int tag);
int wmf_profile_collect(void);
int wmf_profile_entry(int i, char* name, int name_cap, int* tag, double* ms, int64_t* launches, double* min_ms, double* max_ms);
int wmf_profile_reset(void);
/* Kernel-SELECTION switches for timing experiments (tools/kernel_lab.py); default 0, process-wide, not synchronised
 * with running solves.  Every selection computes the same results (the parity suite runs under each of them):
 *       64 plain 32 x 32 Gauss-Jordan for rows with 17..32 entries, 256 no border column, 1024 run-time-indexed eight-wave
 *       kernel for f > 144, 2048 no two-rows-per-wave kernel, 4096 register-ring heavy kernel at k = 128 (instead of the
 *       LDS-DMA ring), 8192 f32 MFMA accumulation in the LDS-DMA kernel, 65536 register-ring heavy kernel at k = 64 (instead of the LDS-DMA ring),
 *       131072 f32 Gramian and 262144 f32 row transform for f = 97 .. 144, 524288 f32 S tiles for rows with <= 32 entries,
 *       2097152 f32 MFMA kernel for f > 144 (it does not split rows above 4096 entries), 16777216 the k = 128 heavy-row
 *       kernel with 16-entry groups at one wave per SIMD (instead of 8-entry groups at two), 268435456 no matrix-free
 *       iteration kernel (csrc/wmf_iter.hip): every row above 32 entries is eliminated, as in round 3, 536870912 the VALU forms
 *       of the float64 Gramian and row transform (csrc/wmf_f64.hip) instead of the v_mfma_f64_16x16x4_f64 ones, 1073741824
 *       wmf_rolled_layout_supported() answers 0 (callers then keep the plain split layout).
 * The ablation switches 1 / 2 / 8 (no elimination / no accumulation MFMAs / no tile inverse: results WRONG) exist only
 * in a -DWMF_LAB build; the shipped library returns WMF_EINVAL for them. */
int wmf_debug_set_flags(int flags);

#ifdef __cplusplus
}
#endif
#endif /* WMF_HIP_H */
