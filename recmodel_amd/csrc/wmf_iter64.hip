// The matrix-free iteration of wmf_iter.hip in FLOAT64, for the reference's cores > 1 variants (RecModel/wmf_model.py:242-309:
// float64 rows; wmf_f64.hip is their device path).  Same algebra in whitened coordinates, V = Y~ R^-1 with G = R^T R:
//
//   (I + E) g = b,  E = V_u^T D V_u,  b = V_u^T p,  ||E|| <= tr E = sum_e |w_e| |v_e|^2 =: tau
//   g = b - E b + E^2 b - ...        stopped when the NEXT term is below one float64 ulp of the answer: |E^k b| tau <= 2^-52 |b|
//
// Each term costs two passes over the gathered rows (2 d f multiply-adds) where the direct kernel forms and factors the f x f
// system (d f^2 / 2 + f^3 / 6): at f = 64, d = 200 (BASELINE.json configs[1], item side) eight terms are a third of the
// arithmetic, and none of it is a dependent 4 x 4 block step.  Rows the bound does not cover (tau above IT64_TAU, any
// negative weight, no convergence in IT64_KMAX applications) are simply NOT marked done: solve64v2_kernel (wmf_f64.hip) then
// factors them as before.
//
// Geometry (as in wmf_iter.hip): NW waves per row; entry e = 4 NW s + 4 w + q sits in slot s of wave w, lane group q, whose 16
// lanes r hold FPD consecutive doubles each; NW = 4 for rows of 33 .. 4 NW NS entries, NW = 1 (four independent rows per
// workgroup) for rows of 1 .. 32 entries.
#include <type_traits>
#include "wmf_common.h"
#include "wmf_internal.h"

#ifndef IT64_TAU
#define IT64_TAU 0.5
#endif
#ifndef IT64_GS
#define IT64_GS 2                                     /* slots of an application worked on together (4: 8 % slower, 1: the same) */
#endif
#ifndef IT64_BETA
#define IT64_BETA 1.5                                 /* the interval's width in units of the probe |E b| / |b| */
#endif
#ifndef IT64_CHEB
#define IT64_CHEB 1                                   /* 0: the plain series at every width */
#endif
#ifndef IT64_KMAX
#define IT64_KMAX 40
#endif

// a double moved across lanes as its two words: the DPP controls of wmf_row16_sum, the lane swaps of wmf_qsum (hipcc turns
// __shfl_xor of a double into LDS permutes)
template <int CTRL>
__device__ __forceinline__ double i64_dpp(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
// sum over the 16 lanes of a group; every lane of the group gets the total (the same butterfly on every lane: same bits)
__device__ __forceinline__ double i64_row16_sum(double v) {
    v += i64_dpp<0xB1>(v);     // quad_perm [1,0,3,2]
    v += i64_dpp<0x4E>(v);     // quad_perm [2,3,0,1]
    v += i64_dpp<0x141>(v);    // row_half_mirror
    v += i64_dpp<0x140>(v);    // row_mirror
    return v;
}
// sum over the four 16-lane groups, lane for lane, in every group (a fixed order)
__device__ __forceinline__ double i64_qsum(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = (int)b, hi = (int)(b >> 32);
    const auto sl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false), sh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const double t = __builtin_bit_cast(double, ((long long)(int)sh[0] << 32) | (unsigned)(int)sl[0]) +
                     __builtin_bit_cast(double, ((long long)(int)sh[1] << 32) | (unsigned)(int)sl[1]);      // {g0+g2, g1+g3, g0+g2, g1+g3}
    const long long c = __builtin_bit_cast(long long, t);
    const int lo2 = (int)c, hi2 = (int)(c >> 32);
    const auto ul = __builtin_amdgcn_permlane16_swap(lo2, lo2, false, false), uh = __builtin_amdgcn_permlane16_swap(hi2, hi2, false, false);
    return __builtin_bit_cast(double, ((long long)(int)uh[0] << 32) | (unsigned)(int)ul[0]) +
           __builtin_bit_cast(double, ((long long)(int)uh[1] << 32) | (unsigned)(int)ul[1]);
}

#ifndef IT64_WIDE_WG
#define IT64_WIDE_WG 2                                /* workgroups per CU asked of the compiler for the f > 128 four-wave form */
#endif
template <int NW, int FPD, int NS>
__global__ __launch_bounds__(256, (NW == 4 && FPD == 9) ? IT64_WIDE_WG : 1) void solve64it_kernel(const double* __restrict__ V, const double* __restrict__ Y, int f, int bias,
                                                        const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                        const double* __restrict__ vals, int64_t n, int dlo, int dhi,
                                                        double* __restrict__ gout, int32_t* __restrict__ state,
                                                        const int32_t* __restrict__ ctrl) {
    constexpr int EPS = 4 * NW, TEAMS = 4 / NW;                 // entries per slot; rows a workgroup works on at once
    constexpr int FEAT = 16 * FPD;
    constexpr bool CHEB = IT64_CHEB && FPD <= 5;                  // (the wider forms have no registers for a third vector)
    __shared__ double part[2][4][FEAT];                          // cross-wave partial sums (NW = 4), two buffers
    __shared__ double psc[2][4][4];
    if (ctrl[0] == 0) return;                                    // no whitened factors this half step (wmf_f64.hip)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wv = NW == 4 ? wave : 0;                           // wave within the row's team
    const int team = NW == 4 ? 0 : wave;
    const int r = lane & 15, q = lane >> 4;
    bool fin[FPD];
#pragma unroll
    for (int j = 0; j < FPD; ++j) fin[j] = r * FPD + j < f;
    int parity = 0;

    for (int64_t row = (int64_t)blockIdx.x * TEAMS + team; row < n; row += (int64_t)gridDim.x * TEAMS) {
        const int64_t lo = indptr[row];
        const int d = (int)min(indptr[row + 1] - lo, (int64_t)(dhi + 1));
        if (d < dlo || d > dhi) continue;                        // (wave uniform: all waves of a team see the same row)
        const int ns = (d + EPS - 1) / EPS;
        // ---- gather: REQUESTS ONLY (nothing here reads what it loads: no wait separates the slots' requests; wmf_iter.hip);
        // 16-byte pieces where the rows allow (f and FPD even), the ids first
        const bool wide = (FPD % 2 == 0) && (f % 2 == 0);
        double vb[NS][FPD], wt[NS], bs[NS];
        int idx[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int e = EPS * s + 4 * wv + q;
            idx[s] = indices[lo + (e < d ? e : 0)];
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s < ns) {
                const int e = EPS * s + 4 * wv + q;
                const double* vrow = V + (int64_t)idx[s] * f + r * FPD;
                if (wide) {
#pragma unroll
                    for (int j = 0; j < FPD / 2; ++j) {
                        const double2 p2 = *reinterpret_cast<const double2*>(fin[2 * j] ? vrow + 2 * j : V);
                        vb[s][2 * j] = p2.x; vb[s][2 * j + 1] = p2.y;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < FPD; ++j) vb[s][j] = *(fin[j] ? vrow + j : V);
                }
                wt[s] = vals[lo + (e < d ? e : 0)];
                bs[s] = bias ? Y[(int64_t)idx[s] * f] : 0.0;     // data - bias[idx] (wmf_model.py:279), subtracted in pass 0
            } else {
#pragma unroll
                for (int j = 0; j < FPD; ++j) vb[s][j] = 0.0;
                wt[s] = 0.0; bs[s] = 0.0;
            }
        }
        bool neg = false;
        // ---- totals over the row: z (per-lane partials over this lane's entries) -> sums over the lane groups and the waves
        auto exchange = [&](double (&z)[FPD], double& s1, double& s2, auto scalars) {       // scalars: pass 0 only (tau, negatives)
            constexpr bool SC = decltype(scalars)::value;
#pragma unroll
            for (int j = 0; j < FPD; ++j) z[j] = i64_qsum(z[j]);
            if constexpr (SC) { s1 = i64_qsum(s1); s2 = i64_qsum(s2); }
            if constexpr (NW == 4) {
                if (q == 0) {
#pragma unroll
                    for (int j = 0; j < FPD; ++j) part[parity][wv][r * FPD + j] = z[j];
                    if constexpr (SC) { if (r == 0) { psc[parity][wv][0] = s1; psc[parity][wv][1] = s2; } }
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < FPD; ++j)
                    z[j] = (part[parity][0][r * FPD + j] + part[parity][1][r * FPD + j]) + (part[parity][2][r * FPD + j] + part[parity][3][r * FPD + j]);
                if constexpr (SC) {
                    s1 = (psc[parity][0][0] + psc[parity][1][0]) + (psc[parity][2][0] + psc[parity][3][0]);
                    s2 = (psc[parity][0][1] + psc[parity][1][1]) + (psc[parity][2][1] + psc[parity][3][1]);
                }
                parity ^= 1;
            }
        };
        auto norm2 = [&](const double (&v)[FPD]) {
            double a = 0.0;
#pragma unroll
            for (int j = 0; j < FPD; ++j) a = __builtin_fma(v[j], v[j], a);
            return i64_row16_sum(a);
        };
        // ---- pass 0: the weights, b = V_u^T p, tau = sum |w| |v|^2, any negative weight
        double bv[FPD], tau = 0.0;
#pragma unroll
        for (int j = 0; j < FPD; ++j) bv[j] = 0.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s < ns) {
                const bool valid = EPS * s + 4 * wv + q < d;
                const double w = wt[s] - bs[s];
                if (valid && !(w >= 0.0)) neg = true;
                wt[s] = valid ? w : 0.0;
#pragma unroll
                for (int j = 0; j < FPD; ++j) vb[s][j] = fin[j] ? vb[s][j] : 0.0;
                const double pp = valid ? w + 1.0 : 0.0;         // p = w + 1 (wmf_model.py:285-287)
                double n2 = 0.0;
#pragma unroll
                for (int j = 0; j < FPD; ++j) {
                    bv[j] = __builtin_fma(pp, vb[s][j], bv[j]);
                    n2 = __builtin_fma(vb[s][j], vb[s][j], n2);
                }
                tau = __builtin_fma(fabs(wt[s]), n2, tau);
            }
        }
        double negs = neg ? 1.0 : 0.0;
        tau = i64_row16_sum(tau);
        negs = i64_row16_sum(negs);
        exchange(bv, tau, negs, std::true_type{});
        const double nb = norm2(bv);
        // (wave-uniform by construction: every lane holds the same totals)
        bool go = tau <= IT64_TAU && negs == 0.0;
        bool converged = false;
        double xv[FPD], yv[FPD];
#pragma unroll
        for (int j = 0; j < FPD; ++j) { xv[j] = bv[j]; yv[j] = bv[j]; }
        // z = E y: pass A (the entries' dots with y, summed over the 16 lanes of a group), pass B, the totals over groups and waves
        auto apply = [&](const double (&y)[FPD], double (&z)[FPD]) {
            double u1 = 0.0, u2 = 0.0;
#pragma unroll
            for (int j = 0; j < FPD; ++j) z[j] = 0.0;
#pragma unroll
            for (int s2 = 0; s2 < NS; s2 += IT64_GS) {               // IT64_GS slots at a time (independent chains); a group past the row's
                if (s2 < ns) {                                       // end is skipped, the empty slots of the last one hold zeros
                    double a[IT64_GS];
#pragma unroll
                    for (int i = 0; i < IT64_GS; ++i) {
                        a[i] = 0.0;
                        if (s2 + i < NS) {
#pragma unroll
                            for (int j = 0; j < FPD; ++j) a[i] = __builtin_fma(vb[s2 + i][j], y[j], a[i]);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < IT64_GS; ++i) a[i] = i64_row16_sum(a[i]);
#pragma unroll
                    for (int i = 0; i < IT64_GS; ++i) {
                        if (s2 + i < NS) {
                            const double t = a[i] * wt[s2 + i];
#pragma unroll
                            for (int j = 0; j < FPD; ++j) z[j] = __builtin_fma(t, vb[s2 + i][j], z[j]);
                        }
                    }
                }
            }
            exchange(z, u1, u2, std::false_type{});
        };
        if (go) {
            const double stop = 4.9e-32 * nb;                    // (2^-52)^2 |b|^2
            double z[FPD];
            apply(yv, z);                                        // the first term of the series, and the probe of ||E||
            double nr = norm2(z);
            if (nr * tau * tau <= stop) {                        // x = b - E b: the next term would be <= tau |z|
                converged = true;
#pragma unroll
                for (int j = 0; j < FPD; ++j) xv[j] -= z[j];
            } else if constexpr (CHEB) {
                // Chebyshev recurrence for (I + E) x = b on [1, 1 + beta] (Saad, Iterative Methods, alg. 12.1), beta = 1.5 |E b| / |b|
                // capped by tr E: E is positive semi-definite here (no negative weight) and b = V_u^T p leans on its dominant
                // directions, so the probe is close to ||E|| from below.  An optimistic beta is safe: for an eigenvalue lambda
                // beyond it the error still contracts while lambda < 2 + beta, and tr E <= IT64_TAU bounds them all; x, r are
                // updated with the same d, so r stays the residual of x whatever the coefficients are (rcp is approximate).
                const double rho1 = __builtin_sqrt(nr / nb);
                const double beta = fmin(tau, fmax(IT64_BETA * rho1, 1e-9));
                const double theta = 1.0 + 0.5 * beta, idelta = 2.0 / beta, sigma = theta * idelta;
                double rho = 1.0 / sigma, rv[FPD], dv[FPD];
                const double itheta = 1.0 / theta;
#pragma unroll
                for (int j = 0; j < FPD; ++j) { rv[j] = -z[j]; dv[j] = rv[j] * itheta; }       // x0 = b, r0 = b - (b + E b)
                for (int k = 1; k < IT64_KMAX; ++k) {
#pragma unroll
                    for (int j = 0; j < FPD; ++j) xv[j] += dv[j];
                    apply(dv, z);
#pragma unroll
                    for (int j = 0; j < FPD; ++j) rv[j] -= dv[j] + z[j];
                    nr = norm2(rv);
                    if (nr <= 16.0 * stop) { converged = true; break; }        // |x - x*| <= |r| (the matrix is >= I): four ulps of |b|
                    if (!(nr == nr)) break;                      // NaN: leave the row to the direct kernel
                    const double rho_n = __builtin_amdgcn_rcp(2.0 * sigma - rho);
                    const double c1 = rho_n * rho, c2 = 2.0 * rho_n * idelta;
#pragma unroll
                    for (int j = 0; j < FPD; ++j) dv[j] = __builtin_fma(c1, dv[j], c2 * rv[j]);
                    rho = rho_n;
                }
            } else {
                double sign = 1.0;
#pragma unroll
                for (int j = 0; j < FPD; ++j) { xv[j] -= z[j]; yv[j] = z[j]; }
                for (int k = 1; k < IT64_KMAX; ++k) {
                    apply(yv, z);
                    nr = norm2(z);
#pragma unroll
                    for (int j = 0; j < FPD; ++j) { xv[j] = __builtin_fma(sign, z[j], xv[j]); yv[j] = z[j]; }
                    if (nr * tau * tau <= stop) { converged = true; break; }  // the next term would be <= tau |z|
                    if (!(nr == nr)) break;
                    sign = -sign;
                }
            }
        }
        if (converged) {
            if (wv == 0 && q == 0) {
#pragma unroll
                for (int j = 0; j < FPD; ++j)
                    if (fin[j]) gout[row * f + r * FPD + j] = xv[j];
                if (r == 0) state[row] = 1;
            }
        }
    }
}

template <int NW, int FPD, int NS>
static void i64_launch(const double* V, const double* Y, int f, int bias, const int64_t* indptr, const int32_t* indices, const double* vals,
                       int64_t n, int dlo, int dhi, double* gout, int32_t* state, const int32_t* ctrl, hipStream_t st) {
    static const char* nm = wmf_kname("solve64it_kernel<%d, %d, %d>", NW, FPD, NS);
    constexpr int TEAMS = 4 / NW;
    int64_t grid = (n + TEAMS - 1) / TEAMS;
    if (grid > 4096) grid = 4096;
    WMF_LAUNCH(nm, (solve64it_kernel<NW, FPD, NS>), dim3((unsigned)grid), dim3(256), 0, st, V, Y, f, bias, indptr, indices, vals, n, dlo, dhi,
               gout, state, ctrl);
}

// longest row the four-wave form holds at width f (0: no kernel)
int wmf_iter64_dmax(int f) {
    if (f <= 64) return 16 * 16;
    if (f <= 80) return 16 * 12;
    if (f <= 144) return 16 * 9;
    return 0;
}

// rows with dlo .. dhi stored entries of a half step in float64: g (whitened coordinates) and state[row] = 1 for the rows solved
int wmf_launch_iter64(const double* V, const double* Y, int f, int bias, const int64_t* indptr, const int32_t* indices,
                      const double* vals, int64_t n, int low, double* gout, int32_t* state, const int32_t* ctrl, hipStream_t st) {
    if (n <= 0) return 0;
    const int dmax = wmf_iter64_dmax(f);
    if (!dmax) return 0;
#define GO(NW, FPD, NS, LO, HI) i64_launch<NW, FPD, NS>(V, Y, f, bias, indptr, indices, vals, n, LO, HI, gout, state, ctrl, st)
    if (low) {                                                   // rows of 1 .. 32 entries: one wave per row
        if (f <= 64) GO(1, 4, 8, 1, 32);
        else if (f <= 80) GO(1, 5, 8, 1, 32);
        else if (f <= 128) GO(1, 8, 8, 1, 32);
        else GO(1, 9, 8, 1, 32);
    } else {
        if (f <= 64) GO(4, 4, 16, 33, 256);
        else if (f <= 80) GO(4, 5, 12, 33, 192);
        else if (f <= 128) GO(4, 8, 9, 33, 144);
        else GO(4, 9, 9, 33, 144);
    }
#undef GO
    return 0;
}
