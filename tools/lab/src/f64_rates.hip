// Lab: issue rates of the float64 instructions the f64 row kernels are built from (one wave per SIMD, then two), gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(double* out, int iters, double seed) {
    double a = seed + threadIdx.x, b = 1.0;
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double x0 = a, x1 = a + 1, x2 = a + 2, x3 = a + 3, x4 = a + 4, x5 = a + 5, x6 = a + 6, x7 = a + 7;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {                // 4 independent MFMAs
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        } else if (MODE == 1) {         // 8 independent FMAs
            x0 = __builtin_fma(x0, b, a); x1 = __builtin_fma(x1, b, a); x2 = __builtin_fma(x2, b, a); x3 = __builtin_fma(x3, b, a);
            x4 = __builtin_fma(x4, b, a); x5 = __builtin_fma(x5, b, a); x6 = __builtin_fma(x6, b, a); x7 = __builtin_fma(x7, b, a);
        } else if (MODE == 2) {         // 4 MFMAs with zero C (the kernels' form) into separate results, summed by FMAs
            d4 z = {0, 0, 0, 0};
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, b, z, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, b, z, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x2, b, z, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x3, b, z, 0, 0, 0);
            x0 = c0[0] * 1e-3; x1 = c1[1] * 1e-3; x2 = c2[2] * 1e-3; x3 = c3[3] * 1e-3;
        } else if (MODE == 4) {         // butterfly sum of a double over 16 lanes through ds_swizzle (LDS pipe) + v_add_f64, 4 chains
            double* xs[4] = {&x0, &x1, &x2, &x3};
#pragma unroll
            for (int st = 0; st < 4; ++st) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const long long bb = __builtin_bit_cast(long long, *xs[c]);
                    int lo, hi;
                    if (st == 0) { lo = __builtin_amdgcn_ds_swizzle((int)bb, 0x041F); hi = __builtin_amdgcn_ds_swizzle((int)(bb >> 32), 0x041F); }
                    else if (st == 1) { lo = __builtin_amdgcn_ds_swizzle((int)bb, 0x081F); hi = __builtin_amdgcn_ds_swizzle((int)(bb >> 32), 0x081F); }
                    else if (st == 2) { lo = __builtin_amdgcn_ds_swizzle((int)bb, 0x101F); hi = __builtin_amdgcn_ds_swizzle((int)(bb >> 32), 0x101F); }
                    else { lo = __builtin_amdgcn_ds_swizzle((int)bb, 0x201F); hi = __builtin_amdgcn_ds_swizzle((int)(bb >> 32), 0x201F); }
                    *xs[c] = *xs[c] * 0.25 + __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
                }
            }
        } else if (MODE == 5) {         // the same butterfly through DPP (VALU only), 4 chains
            double* xs[4] = {&x0, &x1, &x2, &x3};
#pragma unroll
            for (int st = 0; st < 4; ++st) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const long long bb = __builtin_bit_cast(long long, *xs[c]);
                    int lo, hi;
                    if (st == 0) { lo = __builtin_amdgcn_update_dpp(0, (int)bb, 0xB1, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, (int)(bb >> 32), 0xB1, 0xF, 0xF, true); }
                    else if (st == 1) { lo = __builtin_amdgcn_update_dpp(0, (int)bb, 0x4E, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, (int)(bb >> 32), 0x4E, 0xF, 0xF, true); }
                    else if (st == 2) { lo = __builtin_amdgcn_update_dpp(0, (int)bb, 0x141, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, (int)(bb >> 32), 0x141, 0xF, 0xF, true); }
                    else { lo = __builtin_amdgcn_update_dpp(0, (int)bb, 0x140, 0xF, 0xF, true); hi = __builtin_amdgcn_update_dpp(0, (int)(bb >> 32), 0x140, 0xF, 0xF, true); }
                    *xs[c] = *xs[c] * 0.25 + __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
                }
            }
        } else if (MODE == 3) {         // 4x4x4 (4 blocks) MFMAs
            x0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, x0, 0, 0, 0);
            x1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, x1, 0, 0, 0);
            x2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, x2, 0, 0, 0);
            x3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, x3, 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int MODE>
static void run(const char* name, int per_iter, int waves_per_simd) {
    double* out; hipMalloc(&out, 8 * 256 * 4096);
    const int iters = 20000, grid = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    rate_kernel<MODE><<<grid, 256>>>(out, 100, 1.0);
    hipEventRecord(e0);
    rate_kernel<MODE><<<grid, 256>>>(out, iters, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // one workgroup of 4 waves per CU x waves_per_simd: each SIMD runs waves_per_simd waves
    const double ns_per_instr = ms * 1e6 / ((double)iters * per_iter * waves_per_simd);
    printf("%-28s waves/SIMD %d  %.2f ns per instruction per SIMD (%.1f clocks at 2.4 GHz)\n", name, waves_per_simd, ns_per_instr, ns_per_instr * 2.4);
    hipFree(out);
}
int main() {
    for (int w = 1; w <= 4; ++w) {
        if (w == 3) continue;
        run<0>("mfma_f64_16x16x4 (acc chain)", 4, w);
        run<2>("mfma_f64_16x16x4 (zero C)", 4, w);
        run<3>("mfma_f64_4x4x4_4b", 4, w);
        run<1>("v_fma_f64", 8, w);
        run<4>("16-lane f64 sum, ds_swizzle", 4, w);       // per sum of one double
        run<5>("16-lane f64 sum, DPP", 4, w);
    }
    return 0;
}
