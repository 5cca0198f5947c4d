// C ABI of libwmf_hip.so (include/wmf_hip.h): argument checks, the degree-binned row plan, and
// the self-contained host-pointer entry point that replaces WMF.recompute_factors[_bias]
// (RecModel/wmf_model.py:213-240, :311-351).
#include "../../include/wmf_hip.h"
#include "wmf_internal.h"

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <set>
#include <string>
#include <vector>

static thread_local char g_err[512] = "";
int wmf_debug_flags = 0;

void wmf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

#define HIP_TRY(call)                                                                            \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            wmf_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return e_ == hipErrorOutOfMemory ? WMF_ENOMEM : WMF_EHIP;                            \
        }                                                                                        \
    } while (0)

static int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        wmf_set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return WMF_EHIP;
    }
    return WMF_OK;
}

static int check_shape(int f, int ld) {
    if (f < 1 || f > WMF_MAX_F) { wmf_set_error("factor width f=%d outside [1, %d]", f, WMF_MAX_F); return WMF_EINVAL; }
    if (ld < f || (ld & 3)) { wmf_set_error("leading dimension ld=%d must be a multiple of 4 and >= f=%d", ld, f); return WMF_EINVAL; }
    if (ld > 272) { wmf_set_error("leading dimension ld=%d too large", ld); return WMF_EINVAL; }
    return WMF_OK;
}

// ---- per-kernel event timing ------------------------------------------------------------------
// Launch sites bracket every kernel with two events on its own stream (WMF_LAUNCH / WmfProfScope) while profiling is on.
// wmf_profile_collect() waits for the events and folds them into a table keyed by (kernel symbol, tag); the tag is
// whatever wmf_profile_set_tag() last set (bench.py: 0 = users half step, 1 = items half step), so the two launches of
// one kernel per iteration are reported separately.  One mutex guards all of it: the entry points may be called from
// several host threads (one per stream).
struct ProfRec { const char* name; int tag; hipEvent_t a, b; };
struct ProfAgg { const char* name; int tag; double ms, min_ms, max_ms; int64_t launches; };
static std::mutex g_prof_mu;
static std::atomic<bool> g_prof_on{false};
static int g_prof_tag = 0;
static std::vector<ProfRec> g_prof;
static std::vector<ProfAgg> g_prof_table;
static thread_local ProfRec g_prof_open;
static thread_local bool g_prof_is_open = false;

const char* wmf_kname(const char* fmt, ...) {
    char buf[192];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    static std::mutex mu;
    static std::set<std::string> names;
    std::lock_guard<std::mutex> lk(mu);
    return names.insert(buf).first->c_str();       // node-based container: the string never moves
}

void wmf_prof_begin(const char* name, hipStream_t st) {
    if (!g_prof_on.load(std::memory_order_relaxed)) return;
    ProfRec r; r.name = name;
    if (hipEventCreate(&r.a) != hipSuccess) return;
    if (hipEventCreate(&r.b) != hipSuccess) { (void)hipEventDestroy(r.a); return; }
    { std::lock_guard<std::mutex> lk(g_prof_mu); r.tag = g_prof_tag; }
    (void)hipEventRecord(r.a, st);
    g_prof_open = r; g_prof_is_open = true;
}
void wmf_prof_end(hipStream_t st) {
    if (!g_prof_is_open) return;
    (void)hipEventRecord(g_prof_open.b, st);
    g_prof_is_open = false;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back(g_prof_open);
}

extern "C" {

int wmf_debug_set_flags(int flags) {
#ifndef WMF_LAB
    // Ablation switches that make results WRONG (1, 2, 8) and the f32-MFMA accumulation of the LDS-DMA kernel (8192) exist only
    // in a -DWMF_LAB build (tools/build_variant.sh)
    if (flags & (1 | 2 | 8 | 8192)) { wmf_set_error("wmf_debug_set_flags: flags 1/2/8/8192 need a -DWMF_LAB build"); return WMF_EINVAL; }
#endif
    wmf_debug_flags = flags;
    return WMF_OK;
}
int wmf_profile_enable(int on) { g_prof_on.store(on != 0); return WMF_OK; }
int wmf_profile_set_tag(int tag) { std::lock_guard<std::mutex> lk(g_prof_mu); g_prof_tag = tag; return WMF_OK; }
int wmf_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.clear();
    g_prof_table.clear();
    return WMF_OK;
}
int wmf_profile_collect(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) {
        float t = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            ProfAgg* hit = nullptr;
            for (auto& e : g_prof_table) if (e.name == r.name && e.tag == r.tag) { hit = &e; break; }
            if (!hit) { g_prof_table.push_back(ProfAgg{r.name, r.tag, 0.0, 1e300, 0.0, 0}); hit = &g_prof_table.back(); }
            hit->ms += t; hit->launches += 1;
            if (t < hit->min_ms) hit->min_ms = t;
            if (t > hit->max_ms) hit->max_ms = t;
        }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    g_prof.clear();
    return (int)g_prof_table.size();
}
int wmf_profile_entry(int i, char* name, int name_cap, int* tag, double* ms, int64_t* launches, double* min_ms, double* max_ms) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (i < 0 || i >= (int)g_prof_table.size()) { wmf_set_error("wmf_profile_entry: index %d outside the table", i); return WMF_EINVAL; }
    const ProfAgg& e = g_prof_table[(size_t)i];
    if (name && name_cap > 0) { strncpy(name, e.name, (size_t)name_cap - 1); name[name_cap - 1] = 0; }
    if (tag) *tag = e.tag;
    if (ms) *ms = e.ms;
    if (launches) *launches = e.launches;
    if (min_ms) *min_ms = e.min_ms;
    if (max_ms) *max_ms = e.max_ms;
    return WMF_OK;
}

const char* wmf_last_error(void) { return g_err; }
int wmf_version(void) { return 100; }
int wmf_ld_for(int f) { return (f + 3) & ~3; }

// ---- workspace layout of gram/factorize: [partials fp32][slices fp64 32 x f x f][two fp64 images fp x (fp + 2), fp = 16 ceil(f / 16)] ----
static int64_t gram_partial_bytes(int f) {
    const int64_t nfb = (f + 15) / 16, nt = nfb * (nfb + 1) / 2;
    return (int64_t)wmf_gram_max_waves(f) * nt * 256 * (int64_t)sizeof(float);
}
static int64_t gram_slices_off(int f) { return (gram_partial_bytes(f) + 255) & ~(int64_t)255; }
static int64_t gram_a_off(int f) { return (gram_slices_off(f) + (int64_t)32 * f * f * 8 + 255) & ~(int64_t)255; }
int64_t wmf_gram_workspace_bytes(int f) {
    if (f < 1 || f > WMF_MAX_F) return 0;
    const int64_t fp = 16 * ((f + 15) / 16);
    return gram_a_off(f) + 2 * fp * (fp + 2) * (int64_t)sizeof(double) + 256;
}

int wmf_gram(const float* Y, int64_t m, int f, int ld, int bias, double* G_sum, void* workspace, void* stream) {
    int rc = check_shape(f, ld);
    if (rc) return rc;
    if (!Y || !G_sum || !workspace || m < 0) { wmf_set_error("wmf_gram: null pointer or negative m"); return WMF_EINVAL; }
    if (wmf_launch_gram(Y, m, f, ld, bias, G_sum, (float*)workspace, (double*)((char*)workspace + gram_slices_off(f)),
                        (hipStream_t)stream)) {
        wmf_set_error("wmf_gram: unsupported f=%d", f);
        return WMF_EINVAL;
    }
    return check_launch("wmf_gram");
}

int wmf_factorize(const double* G_sum, int f, int ld, double lambda, float* W_white, float* W_unwhite, int32_t* info,
                  void* workspace, void* stream) {
    int rc = check_shape(f, ld);
    if (rc) return rc;
    if (!G_sum || !W_white || !W_unwhite || !info || !workspace) { wmf_set_error("wmf_factorize: null pointer"); return WMF_EINVAL; }
    double* gA = (double*)((char*)workspace + gram_a_off(f));
    wmf_launch_factorize(G_sum, f, ld, lambda, W_white, W_unwhite, info, gA, (hipStream_t)stream);
    return check_launch("wmf_factorize");
}

int wmf_row_transform(const float* in, int64_t m, int f, int ld, const float* W, int set_col0_one, float* out,
                      float* col0_out, void* stream) {
    int rc = check_shape(f, ld);
    if (rc) return rc;
    if (!in || !W || !out || m < 0) { wmf_set_error("wmf_row_transform: null pointer or negative m"); return WMF_EINVAL; }
    rc = wmf_launch_transform(in, m, f, ld, W, set_col0_one, out, col0_out, (hipStream_t)stream);
    if (rc) {
        if (rc == -3) wmf_set_error("wmf_row_transform: f=%d, ld=%d with set_col0_one writes the split layout: col0_out (float[2 m]) is required and out may not alias in", f, ld);
        else wmf_set_error("wmf_row_transform: unsupported f=%d", f);
        return WMF_EINVAL;
    }
    return check_launch("wmf_row_transform");
}

// ---- plan ---------------------------------------------------------------------------------------
static int bin_of(int64_t d, int f) {
    if (d <= 16) return WMF_BIN_LOW16;
    if (d <= 32) return WMF_BIN_LOW32;
    return wmf_direct_supported(f) ? WMF_BIN_MFMA : WMF_BIN_GENERAL;
}

int wmf_plan_create(const int64_t* indptr, int64_t n, int f, int bias, wmf_plan** out) {
    if (!indptr || !out || n < 0 || n > 0x7fffffffLL) { wmf_set_error("wmf_plan_create: bad arguments"); return WMF_EINVAL; }
    if (f < 1 || f > WMF_MAX_F) { wmf_set_error("wmf_plan_create: f=%d unsupported", f); return WMF_EINVAL; }
    wmf_plan* p = new wmf_plan();
    memset(p, 0, sizeof(*p));
    p->n = n; p->f = f;
    std::vector<int32_t> order((size_t)n);
    int64_t start[WMF_NBINS + 1] = {0};
    for (int64_t r = 0; r < n; ++r) {
        const int64_t d = indptr[r + 1] - indptr[r];
        if (d < 0) { delete p; wmf_set_error("wmf_plan_create: indptr not monotone at row %lld", (long long)r); return WMF_EINVAL; }
        p->count[bin_of(d, f)]++;
        p->nnz[bin_of(d, f)] += d;
    }
    for (int b = 0; b < WMF_NBINS; ++b) start[b + 1] = start[b] + p->count[b];
    int64_t fill[WMF_NBINS];
    for (int b = 0; b < WMF_NBINS; ++b) fill[b] = start[b];
    // within the bin of rows with more than 32 entries (one wave per row for f <= 144; four waves per row -- the row-split
    // kernel, f <= 257 -- beyond): ordinary rows first, rows with more than WMF_HEAVY_T entries last (split into segments)
    const int hb = wmf_direct_supported(f) ? WMF_BIN_MFMA : WMF_BIN_GENERAL;
    const bool can_split = wmf_direct_supported(f) || wmf_rowsplit_supported(f);
    std::vector<int32_t> heavy;
    // first bin: rows with at most 8 entries first (two of them share a wave), the others behind them
    for (int64_t r = 0; r < n; ++r) {
        const int64_t d = indptr[r + 1] - indptr[r];
        if (d <= 8) { order[(size_t)fill[WMF_BIN_LOW16]++] = (int32_t)r; p->count8++; p->nnz8 += d; }
    }
    // (round 4) ... and, first of all, the rows short enough for the matrix-free iteration kernel (wmf_iter.hip): at most
    // wmf_iter_dmax entries -- what a workgroup of that kernel keeps in registers at this width
    const bool split_layout = bias && wmf_split_layout(f, wmf_ld_for(f));
    const int64_t iter_dmax = wmf_iter_dmax(f, wmf_ld_for(f), split_layout ? 1 : 0);
    p->iter_dmax = (int)iter_dmax;
    std::vector<int32_t> longer;
    for (int64_t r = 0; r < n; ++r) {
        const int64_t d = indptr[r + 1] - indptr[r];
        const int b = bin_of(d, f);
        if (b == WMF_BIN_LOW16 && d <= 8) continue;
        if (b == hb && can_split && d > WMF_HEAVY_T) { heavy.push_back((int32_t)r); continue; }
        if (b == hb && d > iter_dmax) { longer.push_back((int32_t)r); continue; }
        if (b == hb) { p->iter_count++; p->iter_nnz += d; }
        order[(size_t)fill[b]++] = (int32_t)r;
    }
    for (int32_t r : longer) order[(size_t)fill[hb]++] = r;
    std::vector<int64_t> seg_lo;
    std::vector<int32_t> seg_d, seg_first(1, 0);
    for (int32_t r : heavy) {
        order[(size_t)fill[hb]++] = r;
        const int64_t d = indptr[r + 1] - indptr[r];
        p->heavy_nnz += d;
        for (int64_t off = 0; off < d; off += WMF_SEG) {
            seg_lo.push_back(indptr[r] + off);
            seg_d.push_back((int32_t)(d - off < WMF_SEG ? d - off : WMF_SEG));
        }
        seg_first.push_back((int32_t)seg_lo.size());
    }
    p->heavy_count = (int64_t)heavy.size();
    p->seg_total = (int64_t)seg_lo.size();
    const size_t bytes = (size_t)(n > 0 ? n : 1) * sizeof(int32_t);
    hipError_t e = hipMalloc((void**)&p->rows_all, bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&p->fallback_rows, bytes);
    if (e == hipSuccess) e = hipMalloc((void**)&p->fallback_count, 256);
    if (e == hipSuccess && n > 0) e = hipMemcpy(p->rows_all, order.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(p->fallback_count, 0, 256);
    // every buffer a solve may need is allocated here, so that wmf_solve_rows only enqueues work: the bias-adjusted
    // weights of a biased model, the matrix slices of the pivoted fallback for f > 144
    const int64_t nnz_all = p->nnz[0] + p->nnz[1] + p->nnz[2] + p->nnz[3];
    p->bias = bias != 0;
    // (not at the split-layout widths -- k = 16 m with biases, the headline k = 128 among them: there the row kernels take the
    // bias with the gathered row and never touch w_eff; 400 MB per side at cfg3.  The layout decision is latched here.)
    p->split = bias && wmf_split_layout(f, wmf_ld_for(f));
    if (e == hipSuccess && bias && !p->split && nnz_all > 0) e = hipMalloc((void**)&p->w_eff, (size_t)nnz_all * sizeof(float));
    if (e == hipSuccess && f > 144) e = hipMalloc((void**)&p->wide_ws, wmf_wide_lu_workspace_bytes(f));
    if (e == hipSuccess && p->iter_count > 0) e = hipMalloc((void**)&p->iter_bounce_rows, (size_t)2 * p->iter_count * sizeof(int32_t));   // [final list | stage 1's hand-on list]
    if (e == hipSuccess && p->iter_count > 0) {
        // the candidates' bookkeeping as one 16-byte record per row (the iteration kernel reads it with one scalar load)
        const int32_t* cand = order.data() + start[hb];
        std::vector<int32_t> rec((size_t)p->iter_count * 4);
        for (int64_t i = 0; i < p->iter_count; ++i) {
            const int64_t lo = indptr[cand[i]];
            rec[(size_t)4 * i] = (int32_t)(uint32_t)(lo & 0xffffffffLL);
            rec[(size_t)4 * i + 1] = (int32_t)(uint32_t)((uint64_t)lo >> 32);
            rec[(size_t)4 * i + 2] = cand[i];
            rec[(size_t)4 * i + 3] = (int32_t)(indptr[cand[i] + 1] - lo);
        }
        e = hipMalloc((void**)&p->iter_info, rec.size() * sizeof(int32_t));
        if (e == hipSuccess) e = hipMemcpy(p->iter_info, rec.data(), rec.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && p->iter_count > 0) e = hipMalloc((void**)&p->iter_stats, 8 * sizeof(unsigned long long));
    if (e == hipSuccess && p->iter_count > 0) e = hipMemset(p->iter_stats, 0, 8 * sizeof(unsigned long long));
    if (e == hipSuccess && p->heavy_count > 0) {
        const int64_t nfb = (f + 15) / 16, nt = nfb * (nfb + 1) / 2 + nfb;
        const int64_t pfloats = wmf_direct_supported(f) ? nt * 256 : wmf_rowsplit_partial_floats(f);
        e = hipMalloc((void**)&p->seg_lo, seg_lo.size() * sizeof(int64_t));
        if (e == hipSuccess) e = hipMalloc((void**)&p->seg_d, seg_d.size() * sizeof(int32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&p->seg_first, seg_first.size() * sizeof(int32_t));
        if (e == hipSuccess) e = hipMalloc((void**)&p->partial, (size_t)p->seg_total * pfloats * sizeof(float));
        if (e == hipSuccess) e = hipMemcpy(p->seg_lo, seg_lo.data(), seg_lo.size() * sizeof(int64_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(p->seg_d, seg_d.data(), seg_d.size() * sizeof(int32_t), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(p->seg_first, seg_first.data(), seg_first.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    }
    if (e != hipSuccess) {
        wmf_set_error("wmf_plan_create: %s", hipGetErrorString(e));
        wmf_plan_destroy(p);
        return e == hipErrorOutOfMemory ? WMF_ENOMEM : WMF_EHIP;
    }
    for (int b = 0; b < WMF_NBINS; ++b) p->rows[b] = p->rows_all + start[b];
    *out = p;
    return WMF_OK;
}

void wmf_plan_destroy(wmf_plan* p) {
    if (!p) return;
    if (p->rows_all) (void)hipFree(p->rows_all);
    if (p->fallback_rows) (void)hipFree(p->fallback_rows);
    if (p->fallback_count) (void)hipFree(p->fallback_count);
    if (p->w_eff) (void)hipFree(p->w_eff);
    if (p->wide_ws) (void)hipFree(p->wide_ws);
    if (p->iter_bounce_rows) (void)hipFree(p->iter_bounce_rows);
    if (p->iter_stats) (void)hipFree(p->iter_stats);
    if (p->iter_info) (void)hipFree(p->iter_info);
    if (p->seg_lo) (void)hipFree(p->seg_lo);
    if (p->seg_d) (void)hipFree(p->seg_d);
    if (p->seg_first) (void)hipFree(p->seg_first);
    if (p->partial) (void)hipFree(p->partial);
    delete p;
}

int wmf_plan_stats(const wmf_plan* p, int64_t* out8 /* int64[14] */) {
    if (!p || !out8) { wmf_set_error("wmf_plan_stats: null"); return WMF_EINVAL; }
    for (int b = 0; b < WMF_NBINS; ++b) { out8[b] = p->count[b]; out8[WMF_NBINS + b] = p->nnz[b]; }
    out8[8] = p->count8; out8[9] = p->nnz8;
    out8[10] = p->heavy_count; out8[11] = p->heavy_nnz;
    out8[12] = p->iter_count; out8[13] = p->iter_nnz;
    return WMF_OK;
}

// What the iteration kernel (wmf_iter.hip) did with its candidates since the last call: rows solved, rows handed back to the
// elimination kernels, applications of the row operator in total, rows solved by the Chebyshev (rather than Neumann) recurrence.
// Synchronises the device; clears the counters.
int wmf_plan_iter_stats(wmf_plan* p, int64_t* out4) {
    if (!p || !out4) { wmf_set_error("wmf_plan_iter_stats: null"); return WMF_EINVAL; }
    out4[0] = out4[1] = out4[2] = out4[3] = 0;
    if (!p->iter_stats) return WMF_OK;
    unsigned long long h[8];                                     // ([4]: rows the first stage handed to the second, not reported)
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(h, p->iter_stats, sizeof(h), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemset(p->iter_stats, 0, sizeof(h));
    if (e != hipSuccess) { wmf_set_error("wmf_plan_iter_stats: %s", hipGetErrorString(e)); return WMF_EHIP; }
    for (int i = 0; i < 4; ++i) out4[i] = (int64_t)h[i];
    return WMF_OK;
}

int wmf_rolled_layout_supported(int f, int ld) { return (check_shape(f, ld) == 0 && wmf_rolled_layout(f, ld)) ? 1 : 0; }

int wmf_solve_rows(const wmf_plan* plan, const float* V, const float* bias_fixed, const int64_t* indptr,
                   const int32_t* indices, const float* values, int64_t n, int f, int ld, float* g,
                   int32_t* fail_count, void* stream) {
    return wmf_solve_rows_ex(plan, V, bias_fixed, indptr, indices, values, n, f, ld, g, fail_count, 0, stream);
}

int wmf_solve_rows_ex(const wmf_plan* plan, const float* V, const float* bias_fixed, const int64_t* indptr,
                      const int32_t* indices, const float* values, int64_t n, int f, int ld, float* g,
                      int32_t* fail_count, int flags, void* stream) {
    int rc = check_shape(f, ld);
    if (rc) return rc;
    if (flags & ~WMF_SOLVE_ROLLED) { wmf_set_error("wmf_solve_rows_ex: unknown flags %d", flags); return WMF_EINVAL; }
    if ((flags & WMF_SOLVE_ROLLED) && (!bias_fixed || !wmf_rolled_layout(f, ld))) {
        wmf_set_error("wmf_solve_rows_ex: WMF_SOLVE_ROLLED needs bias_fixed and wmf_rolled_layout_supported(f=%d, ld=%d)", f, ld);
        return WMF_EINVAL;
    }
    if (plan) plan->rolled = (flags & WMF_SOLVE_ROLLED) ? 1 : 0;
    if (!plan || !V || !indptr || !g || !fail_count) { wmf_set_error("wmf_solve_rows: null pointer"); return WMF_EINVAL; }
    if (bias_fixed && !plan->bias) { wmf_set_error("wmf_solve_rows: bias_fixed given, but the plan was created with bias = 0"); return WMF_EINVAL; }
    if (plan->n != n || plan->f != f) { wmf_set_error("wmf_solve_rows: plan was built for n=%lld f=%d", (long long)plan->n, plan->f); return WMF_EINVAL; }
    if (n == 0) return WMF_OK;
    const int lrc = wmf_launch_solve(plan, V, bias_fixed, indptr, indices, values, f, ld, g, fail_count, (hipStream_t)stream);
    if (lrc == -2) { wmf_set_error("wmf_solve_rows: hipMemsetAsync failed"); return WMF_EHIP; }
    if (lrc == -3) { wmf_set_error("wmf_solve_rows: the plan was created for the split layout of the whitened factors, the call is not (wmf_debug_set_flags(256) changed in between?)"); return WMF_EINVAL; }
    if (lrc) { wmf_set_error("wmf_solve_rows: no kernel for f=%d, ld=%d", f, ld); return WMF_EINVAL; }
    return check_launch("wmf_solve_rows");
}

int64_t wmf_eval_workspace_bytes(void) { return (int64_t)WMF_EVAL_MAX_BLOCKS * 3 * sizeof(double); }

int wmf_eval_sqerr(const float* users, const float* items, int f, int ld, int bias, const int64_t* indptr,
                   const int32_t* indices, const float* values, int64_t n, double* out3, void* workspace, void* stream) {
    int rc = check_shape(f, ld);
    if (rc) return rc;
    if (!users || !items || !indptr || !out3 || !workspace || n < 0) { wmf_set_error("wmf_eval_sqerr: null pointer"); return WMF_EINVAL; }
    wmf_launch_eval(users, items, f, ld, bias, indptr, indices, values, n, out3, (double*)workspace, (hipStream_t)stream);
    return check_launch("wmf_eval_sqerr");
}

int wmf_predict_pairs(const float* users, const float* items, int f, int ld, int bias, const int32_t* users_idx,
                      int64_t n_u, const int32_t* items_idx, int64_t n_i, float* out, void* stream) {
    int rc = check_shape(f, ld);
    if (rc) return rc;
    if (!users || !items || !users_idx || !items_idx || !out) { wmf_set_error("wmf_predict_pairs: null pointer"); return WMF_EINVAL; }
    if (n_u != n_i && n_u != 1 && n_i != 1) {
        wmf_set_error("users and items need to have the same length or only one user / item needs to be provided.");
        return WMF_EINVAL;
    }
    wmf_launch_predict(users, items, f, ld, bias, users_idx, n_u, items_idx, n_i, out, (hipStream_t)stream);
    return check_launch("wmf_predict_pairs");
}

int64_t wmf_partial_row_floats(int f) { return wmf_directw_partial_floats(f); }

int wmf_whitened_row_floats(int f, int ld, int bias) {
    if (check_shape(f, ld)) return 0;
    return (bias && wmf_split_layout(f, ld)) ? f - 1 : ld;
}

int wmf_accumulate_rows(const float* V, const float* bias_fixed, const int64_t* indptr, const int32_t* degrees,
                        const int32_t* indices, const float* values, int64_t n, int64_t nnz, int f, int ld, float* partial,
                        int32_t slot_stride, int32_t slot_offset, float* w_eff_workspace, void* stream) {
    int rc = check_shape(f, ld);
    if (rc) return rc;
    if (!V || !indptr || !degrees || !partial || n < 0 || nnz < 0 || (nnz > 0 && (!indices || !values)) || slot_stride < 1 ||
        slot_offset < 0 || slot_offset >= slot_stride) {
        wmf_set_error("wmf_accumulate_rows: null pointer, negative size or bad slot"); return WMF_EINVAL;
    }
    const int64_t pr = wmf_directw_partial_floats(f);
    if (pr == 0) { wmf_set_error("wmf_accumulate_rows: f=%d not supported (f <= 144)", f); return WMF_EINVAL; }
    hipStream_t st = (hipStream_t)stream;
    if (n == 0) return WMF_OK;
    if (nnz == 0) {                                               // nothing stored here: every partial system of this slot is zero
        if (hipMemset2DAsync(partial + (size_t)slot_offset * pr, (size_t)slot_stride * pr * sizeof(float), 0, (size_t)pr * sizeof(float),
                             (size_t)n, st) != hipSuccess) { wmf_set_error("wmf_accumulate_rows: memset failed"); return WMF_EHIP; }
        return WMF_OK;
    }
    const float* side = nullptr;
    if (bias_fixed && wmf_split_layout(f, ld)) {
        side = bias_fixed;                                        // split layout: the kernel takes the bias from the pairs
    } else if (bias_fixed) {
        if (!w_eff_workspace) { wmf_set_error("wmf_accumulate_rows: bias_fixed needs w_eff_workspace at f=%d, ld=%d", f, ld); return WMF_EINVAL; }
        wmf_launch_bias_adjust(values, indices, bias_fixed, nnz, w_eff_workspace, st);
        values = w_eff_workspace;
    }
    {
        if (wmf_launch_accumulate(V, side, indptr, degrees, indices, values, n, f, ld, partial, slot_stride, slot_offset, st)) { wmf_set_error("wmf_accumulate_rows: no kernel for f=%d", f); return WMF_EINVAL; }
    }
    return check_launch("wmf_accumulate_rows");
}

int wmf_eliminate_rows(float* partial, int64_t n, int32_t slots_per_row, int f, int ld, float* g, int32_t* fail_count,
                       int32_t* scratch, void* stream) {
    int rc = check_shape(f, ld);
    if (rc) return rc;
    if (!partial || !g || !fail_count || !scratch || n < 0 || slots_per_row < 1) { wmf_set_error("wmf_eliminate_rows: null pointer or bad size"); return WMF_EINVAL; }
    if (wmf_directw_partial_floats(f) == 0) { wmf_set_error("wmf_eliminate_rows: f=%d not supported (f <= 144)", f); return WMF_EINVAL; }
    if (n == 0) return WMF_OK;
    {
        if (wmf_launch_eliminate(partial, n, slots_per_row, f, ld, g, scratch, fail_count, (hipStream_t)stream)) { wmf_set_error("wmf_eliminate_rows: no kernel for f=%d", f); return WMF_EINVAL; }
    }
    return check_launch("wmf_eliminate_rows");
}

int64_t wmf_rank_workspace_bytes(int64_t n_cand) { return wmf_rank_ws_bytes(n_cand); }

int wmf_rank_topn(const float* users, const float* items, int f, int ld, int bias, const int32_t* user_idx,
                  const int32_t* cand_idx, int64_t n_cand, int64_t topn, int32_t* out_pos, float* out_scores,
                  void* workspace, int64_t workspace_bytes, void* stream) {
    int rc = check_shape(f, ld);
    if (rc) return rc;
    if (!users || !items || !user_idx || !cand_idx || !out_pos || !workspace) { wmf_set_error("wmf_rank_topn: null pointer"); return WMF_EINVAL; }
    if (n_cand < 1 || n_cand > 0x7fffffffLL || topn < 1 || topn > n_cand) {
        wmf_set_error("wmf_rank_topn: need 1 <= topn <= n_cand < 2^31 (topn=%lld, n_cand=%lld)", (long long)topn, (long long)n_cand);
        return WMF_EINVAL;
    }
    const int lrc = wmf_launch_rank(users, items, f, ld, bias, user_idx, cand_idx, n_cand, topn, out_pos, out_scores, workspace,
                                    workspace_bytes, (hipStream_t)stream);
    if (lrc == -3) { wmf_set_error("wmf_rank_topn: workspace too small (%lld < %lld bytes)", (long long)workspace_bytes, (long long)wmf_rank_ws_bytes(n_cand)); return WMF_EINVAL; }
    if (lrc == -4) { wmf_set_error("wmf_rank_topn: too many keys for the device sort (2^32 or more)"); return WMF_EINVAL; }
    if (lrc) { wmf_set_error("wmf_rank_topn: device sort or copy failed"); return WMF_EHIP; }
    return check_launch("wmf_rank_topn");
}

int64_t wmf_rank_batch_workspace_bytes(int64_t n_users, int64_t n_cand) { return wmf_rank_batch_ws_bytes(n_users, n_cand); }

int wmf_rank_topn_batch(const float* users, const float* items, int f, int ld, int bias, const int32_t* user_idx, int64_t n_users,
                        const int32_t* cand_idx, int64_t n_cand, int64_t topn, int32_t* out_pos, float* out_scores,
                        void* workspace, int64_t workspace_bytes, void* stream) {
    int rc = check_shape(f, ld);
    if (rc) return rc;
    if (!users || !items || !user_idx || !cand_idx || !out_pos || !workspace) { wmf_set_error("wmf_rank_topn_batch: null pointer"); return WMF_EINVAL; }
    if (n_users < 1 || n_cand < 1 || topn < 1 || topn > n_cand || n_users * n_cand >= ((int64_t)1 << 31)) {
        wmf_set_error("wmf_rank_topn_batch: need n_users, n_cand >= 1, 1 <= topn <= n_cand, n_users * n_cand < 2^31");
        return WMF_EINVAL;
    }
    const int lrc = wmf_launch_rank_batch(users, items, f, ld, bias, user_idx, n_users, cand_idx, n_cand, topn, out_pos, out_scores,
                                          workspace, workspace_bytes, (hipStream_t)stream);
    if (lrc == -3) { wmf_set_error("wmf_rank_topn_batch: workspace too small"); return WMF_EINVAL; }
    if (lrc == -4) { wmf_set_error("wmf_rank_topn_batch: too many keys for the device sort (2^32 or more)"); return WMF_EINVAL; }
    if (lrc) { wmf_set_error("wmf_rank_topn_batch: device sort failed"); return WMF_EHIP; }
    return check_launch("wmf_rank_topn_batch");
}

int wmf_hit_counts(const float* users, const float* items, int f, int ld, int bias, const int32_t* pair_user,
                   const int32_t* pair_item, const int32_t* pair_row, int64_t n_pairs, const int32_t* candidates,
                   int32_t n_cand, const int32_t* slot, const int32_t* topn, int32_t n_topn, int64_t* hits, void* stream) {
    int rc = check_shape(f, ld);
    if (rc) return rc;
    if (!users || !items || !topn || !hits || n_pairs < 0 || n_topn < 1 || n_topn > 64 || n_cand < 1) {
        wmf_set_error("wmf_hit_counts: bad arguments (n_pairs=%lld, n_cand=%d, n_topn=%d)", (long long)n_pairs, n_cand, n_topn);
        return WMF_EINVAL;
    }
    if (n_pairs > 0 && (!pair_user || !pair_item || !pair_row || !candidates || !slot)) { wmf_set_error("wmf_hit_counts: null pointer"); return WMF_EINVAL; }
    if (wmf_launch_hits(users, items, ld, bias, pair_user, pair_item, pair_row, n_pairs, candidates, n_cand, slot, topn, n_topn,
                        hits, (hipStream_t)stream)) { wmf_set_error("wmf_hit_counts: hipMemsetAsync failed"); return WMF_EHIP; }
    return check_launch("wmf_hit_counts");
}

int wmf_gather_rows(const float* in, int ld, const int64_t* rows, int64_t n, float* out, void* stream) {
    if (n < 0 || ld < 4 || (ld & 3) || (n > 0 && (!in || !rows || !out))) { wmf_set_error("wmf_gather_rows: null pointer or bad size"); return WMF_EINVAL; }
    wmf_launch_gather_rows(in, ld, rows, n, out, (hipStream_t)stream);
    return check_launch("wmf_gather_rows");
}

int wmf_spmm_rows(const float* V, const int64_t* indptr, const int32_t* indices, const float* values, int64_t n, int f,
                  int ld, float* g, void* stream) {
    int rc = check_shape(f, ld);
    if (rc) return rc;
    if (!V || !indptr || !g || n < 0) { wmf_set_error("wmf_spmm_rows: null pointer"); return WMF_EINVAL; }
    wmf_launch_spmm(V, indptr, indices, values, n, ld, g, (hipStream_t)stream);
    return check_launch("wmf_spmm_rows");
}

int64_t wmf_coo_to_csr_workspace_bytes(int64_t nnz, int64_t n_rows, int64_t n_cols) {
    if (nnz < 0 || n_rows < 0 || n_cols < 1) return 0;
    return wmf_csr_ws_bytes(nnz, n_rows, n_cols);
}

int wmf_coo_to_csr(const int64_t* rows, const int64_t* cols, const float* values, int64_t nnz, int64_t n_rows, int64_t n_cols,
                   int64_t* indptr_out, int32_t* indices_out, float* values_out, int32_t* bad_flag, void* workspace,
                   int64_t workspace_bytes, void* stream) {
    if (nnz < 0 || n_rows < 0 || n_cols < 1 || n_cols > 0x7fffffffLL || !indptr_out || !bad_flag ||
        (nnz > 0 && (!rows || !cols || !values || !indices_out || !values_out || !workspace))) {
        wmf_set_error("wmf_coo_to_csr: null pointer or bad size"); return WMF_EINVAL;
    }
    const int rc = wmf_launch_coo_to_csr(rows, cols, values, nnz, n_rows, n_cols, indptr_out, indices_out, values_out, bad_flag,
                                         workspace, workspace_bytes, (hipStream_t)stream);
    if (rc == -3) { wmf_set_error("wmf_coo_to_csr: workspace too small (need %lld bytes)", (long long)wmf_csr_ws_bytes(nnz, n_rows, n_cols)); return WMF_EINVAL; }
    if (rc == -4) { wmf_set_error("wmf_coo_to_csr: n_rows * n_cols must fit 63 bits and nnz 32"); return WMF_EINVAL; }
    if (rc) { wmf_set_error("wmf_coo_to_csr: sort failed"); return WMF_EHIP; }
    return check_launch("wmf_coo_to_csr");
}

int wmf_confidence_transform(float* values, int64_t nnz, double alpha, double beta, int mode, void* stream) {
    if ((!values && nnz > 0) || nnz < 0 || (mode != 0 && mode != 1)) { wmf_set_error("wmf_confidence_transform: bad arguments"); return WMF_EINVAL; }
    wmf_launch_confidence(values, nnz, alpha, beta, mode, (hipStream_t)stream);
    return check_launch("wmf_confidence_transform");
}

int wmf_confidence_transform_f64(double* values, int64_t nnz, double alpha, double beta, int mode, void* stream) {
    if ((!values && nnz > 0) || nnz < 0 || (mode != 0 && mode != 1)) { wmf_set_error("wmf_confidence_transform_f64: bad arguments"); return WMF_EINVAL; }
    wmf_launch_confidence_f64(values, nnz, alpha, beta, mode, (hipStream_t)stream);
    return check_launch("wmf_confidence_transform_f64");
}

// ---- float64 half step (the reference's cores > 1 variants) ---------------------------------------
int64_t wmf_half_step_f64_workspace_bytes(int f, int64_t m, int64_t n) {
    if (f < 1 || f > WMF_MAX_F || m < 0 || n < 0) return 0;
    return wmf_f64_ws_bytes(f, m, n);
}

int wmf_half_step_f64(const double* Y, int64_t m, int f, int bias, const int64_t* indptr, const int32_t* indices,
                      const double* values, int64_t n, double lambda, double* X, void* workspace, int64_t workspace_bytes,
                      int32_t* fail_count, void* stream) {
    if (f < 1 || f > WMF_MAX_F) { wmf_set_error("wmf_half_step_f64: f=%d unsupported", f); return WMF_EINVAL; }
    if (!Y || !indptr || !workspace || !fail_count || m < 1 || n < 0 || (n > 0 && !X)) {
        wmf_set_error("wmf_half_step_f64: null pointer or bad size"); return WMF_EINVAL;
    }
    if (workspace_bytes < wmf_f64_ws_bytes(f, m, n)) {
        wmf_set_error("wmf_half_step_f64: workspace too small (need %lld bytes)", (long long)wmf_f64_ws_bytes(f, m, n)); return WMF_EINVAL;
    }
    wmf_launch_half_step_f64(Y, m, f, bias, indptr, indices, values, n, lambda, X, workspace, fail_count, (hipStream_t)stream);
    return check_launch("wmf_half_step_f64");
}

// ---- host-level drop-in -------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16) == hipSuccess ? 0 : -1; }
};

int wmf_recompute_factors_f64_host(const double* Y_host, int64_t m, int f, int bias, const int64_t* indptr,
                                   const int32_t* indices, const double* values, int64_t n, double lambda, double* X_host) {
    if (!Y_host || !indptr || !X_host || m < 1 || n < 0 || f < 1 || f > WMF_MAX_F) {
        wmf_set_error("wmf_recompute_factors_f64_host: bad arguments");
        return WMF_EINVAL;
    }
    const int64_t nnz = indptr[n];
    if (nnz > 0 && (!indices || !values)) { wmf_set_error("wmf_recompute_factors_f64_host: null CSR arrays"); return WMF_EINVAL; }
    for (int64_t r = 0; r < n; ++r)
        if (indptr[r + 1] < indptr[r]) { wmf_set_error("wmf_recompute_factors_f64_host: indptr not monotone at row %lld", (long long)r); return WMF_EINVAL; }
    for (int64_t j = 0; j < nnz; ++j)
        if (indices[j] < 0 || indices[j] >= m) { wmf_set_error("column index %d out of range at entry %lld", indices[j], (long long)j); return WMF_EINVAL; }
    DevBuf dY, dPtr, dIdx, dVal, dX, dWs, dFail;
    const int64_t wsb = wmf_f64_ws_bytes(f, m, n);
    if (dY.alloc((size_t)m * f * 8) || dPtr.alloc((size_t)(n + 1) * 8) || dIdx.alloc((size_t)nnz * 4) || dVal.alloc((size_t)nnz * 8) ||
        dX.alloc((size_t)(n > 0 ? n : 1) * f * 8) || dWs.alloc((size_t)wsb) || dFail.alloc(16)) {
        wmf_set_error("wmf_recompute_factors_f64_host: device allocation failed");
        return WMF_ENOMEM;
    }
    HIP_TRY(hipMemcpy(dY.p, Y_host, (size_t)m * f * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dPtr.p, indptr, (size_t)(n + 1) * 8, hipMemcpyHostToDevice));
    if (nnz > 0) {
        HIP_TRY(hipMemcpy(dIdx.p, indices, (size_t)nnz * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dVal.p, values, (size_t)nnz * 8, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemset(dFail.p, 0, 16));
    hipStream_t st = nullptr;
    const int rc = wmf_half_step_f64((const double*)dY.p, m, f, bias, (const int64_t*)dPtr.p, (const int32_t*)dIdx.p,
                                     (const double*)dVal.p, n, lambda, (double*)dX.p, dWs.p, wsb, (int32_t*)dFail.p, st);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    int32_t fail = 0;
    HIP_TRY(hipMemcpy(&fail, dFail.p, 4, hipMemcpyDeviceToHost));
    if (n > 0) HIP_TRY(hipMemcpy(X_host, dX.p, (size_t)n * f * 8, hipMemcpyDeviceToHost));
    if (fail) { wmf_set_error("%d row systems were singular", fail); return WMF_ENUMERIC; }
    return WMF_OK;
}

int wmf_recompute_factors_host(const float* Y_host, int64_t m, int f, int bias, const int64_t* indptr,
                               const int32_t* indices, const float* values, int64_t n, double lambda, float* X_host) {
    if (!Y_host || !indptr || !X_host || m < 1 || n < 0 || f < 1 || f > WMF_MAX_F) {
        wmf_set_error("wmf_recompute_factors_host: bad arguments (m=%lld, n=%lld, f=%d, Y=%p, indptr=%p, X=%p)", (long long)m, (long long)n, f,
                      (const void*)Y_host, (const void*)indptr, (const void*)X_host);
        return WMF_EINVAL;
    }
    const int ld = wmf_ld_for(f);
    const int64_t nnz = indptr[n];
    if (nnz > 0 && (!indices || !values)) { wmf_set_error("wmf_recompute_factors_host: null CSR arrays"); return WMF_EINVAL; }
    for (int64_t j = 0; j < nnz; ++j)
        if (indices[j] < 0 || indices[j] >= m) { wmf_set_error("column index %d out of range at entry %lld", indices[j], (long long)j); return WMF_EINVAL; }
    DevBuf dY, dV, dG, dWw, dWu, dInfo, dWs, dBias, dPtr, dIdx, dVal, dg, dX, dFail;
    const size_t fac_m = (size_t)m * ld * 4, fac_n = (size_t)(n > 0 ? n : 1) * ld * 4;
    if (dY.alloc(fac_m) || dV.alloc(fac_m) || dG.alloc((size_t)f * f * 8) || dWw.alloc((size_t)f * ld * 4) ||
        dWu.alloc((size_t)f * ld * 4) || dInfo.alloc(16) || dWs.alloc((size_t)wmf_gram_workspace_bytes(f)) ||
        dBias.alloc((size_t)m * 8) || dPtr.alloc((size_t)(n + 1) * 8) || dIdx.alloc((size_t)nnz * 4) ||
        dVal.alloc((size_t)nnz * 4) || dg.alloc(fac_n) || dX.alloc(fac_n) || dFail.alloc(16)) {
        wmf_set_error("wmf_recompute_factors_host: device allocation failed");
        return WMF_ENOMEM;
    }
    HIP_TRY(hipMemset(dY.p, 0, fac_m));
    HIP_TRY(hipMemcpy2D(dY.p, (size_t)ld * 4, Y_host, (size_t)f * 4, (size_t)f * 4, (size_t)m, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dPtr.p, indptr, (size_t)(n + 1) * 8, hipMemcpyHostToDevice));
    if (nnz > 0) {
        HIP_TRY(hipMemcpy(dIdx.p, indices, (size_t)nnz * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(dVal.p, values, (size_t)nnz * 4, hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemset(dFail.p, 0, 16));
    HIP_TRY(hipMemset(dInfo.p, 0, 16));
    wmf_plan* plan = nullptr;
    int rc = wmf_plan_create(indptr, n, f, bias, &plan);
    if (rc) return rc;
    hipStream_t st = nullptr;
    rc = wmf_gram((const float*)dY.p, m, f, ld, bias, (double*)dG.p, dWs.p, st);
    if (!rc) rc = wmf_factorize((const double*)dG.p, f, ld, lambda, (float*)dWw.p, (float*)dWu.p, (int32_t*)dInfo.p, dWs.p, st);
    if (!rc) rc = wmf_row_transform((const float*)dY.p, m, f, ld, (const float*)dWw.p, bias, (float*)dV.p, bias ? (float*)dBias.p : nullptr, st);
    if (!rc) rc = wmf_solve_rows(plan, (const float*)dV.p, bias ? (const float*)dBias.p : nullptr, (const int64_t*)dPtr.p,
                                 (const int32_t*)dIdx.p, (const float*)dVal.p, n, f, ld, (float*)dg.p, (int32_t*)dFail.p, st);
    if (!rc) rc = wmf_row_transform((const float*)dg.p, n, f, ld, (const float*)dWu.p, 0, (float*)dX.p, nullptr, st);
    wmf_plan_destroy(plan);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(st));
    int32_t info = 0, fail = 0;
    HIP_TRY(hipMemcpy(&info, dInfo.p, 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(&fail, dFail.p, 4, hipMemcpyDeviceToHost));
    if (info) { wmf_set_error("Gramian + lambda*I is not positive definite (leading minor %d)", info); return WMF_ENUMERIC; }
    if (n > 0) HIP_TRY(hipMemcpy2D(X_host, (size_t)f * 4, dX.p, (size_t)ld * 4, (size_t)f * 4, (size_t)n, hipMemcpyDeviceToHost));
    if (fail) { wmf_set_error("%d row systems were singular", fail); return WMF_ENUMERIC; }
    return WMF_OK;
}

}  // extern "C"
