// Microbenchmark: do f32 MFMA (16x16x4) and f32 VALU work overlap on gfx950 (a) interleaved inside one wave,
// (b) from different waves of the same SIMD?   hipcc --offload-arch=gfx950 -O3 -o test_overlap test_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// mode 0: MFMA only; 1: VALU only; 2: interleaved in every wave (1 MFMA : 8 VALU)
// mode 3: waves with even wave id do MFMA, odd do VALU (needs >= 2 waves per SIMD: block of 512)
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    float v0 = a, v1 = a + 1, v2 = a + 2, v3 = a + 3, v4 = a + 4, v5 = a + 5, v6 = a + 6, v7 = a + 7;
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 4) == 0);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 4) != 0);
    for (int i = 0; i < iters; ++i) {
        if (do_m) { c0 = MFMA(a, b, c0); }
        if (do_v) { v0 = fmaf(v0, b, a); v1 = fmaf(v1, b, a); v2 = fmaf(v2, b, a); v3 = fmaf(v3, b, a);
                    v4 = fmaf(v4, b, a); v5 = fmaf(v5, b, a); v6 = fmaf(v6, b, a); v7 = fmaf(v7, b, a); }
        if (do_m) { c1 = MFMA(a, b, c1); }
        if (do_v) { v0 = fmaf(v0, b, a); v1 = fmaf(v1, b, a); v2 = fmaf(v2, b, a); v3 = fmaf(v3, b, a);
                    v4 = fmaf(v4, b, a); v5 = fmaf(v5, b, a); v6 = fmaf(v6, b, a); v7 = fmaf(v7, b, a); }
        if (do_m) { c2 = MFMA(a, b, c2); }
        if (do_v) { v0 = fmaf(v0, b, a); v1 = fmaf(v1, b, a); v2 = fmaf(v2, b, a); v3 = fmaf(v3, b, a);
                    v4 = fmaf(v4, b, a); v5 = fmaf(v5, b, a); v6 = fmaf(v6, b, a); v7 = fmaf(v7, b, a); }
        if (do_m) { c3 = MFMA(a, b, c3); }
        if (do_v) { v0 = fmaf(v0, b, a); v1 = fmaf(v1, b, a); v2 = fmaf(v2, b, a); v3 = fmaf(v3, b, a);
                    v4 = fmaf(v4, b, a); v5 = fmaf(v5, b, a); v6 = fmaf(v6, b, a); v7 = fmaf(v7, b, a); }
    }
    out[blockIdx.x * 512 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
}

// clean wave specialisation: waves 0-3 of the block run an MFMA-only loop, waves 4-7 a VALU-only loop (PK: v_pk_fma_f32)
template <int PK>
__global__ __launch_bounds__(512) void k2(float* out, int iters, int which) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float a = threadIdx.x * 1e-3f, b = 1.0001f, res = 0.f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    if ((wave & 4) == 0) {
        if (which & 1) {
            f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
            for (int i = 0; i < iters; ++i) { c0 = MFMA(a, b, c0); c1 = MFMA(a, b, c1); c2 = MFMA(a, b, c2); c3 = MFMA(a, b, c3); }
            res = c0[0] + c1[1] + c2[2] + c3[3];
        }
    } else if (which & 2) {
        if (PK) {
            f32x2 v[8], bb = {b, b}, aa = {a, a};
            for (int j = 0; j < 8; ++j) v[j] = f32x2{a + j, a - j};
            for (int i = 0; i < iters; ++i)
#pragma unroll
                for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = __builtin_elementwise_fma(v[j], bb, aa);
            for (int j = 0; j < 8; ++j) res += v[j][0] + v[j][1];
        } else {
            float v[8];
            for (int j = 0; j < 8; ++j) v[j] = a + j;
            for (int i = 0; i < iters; ++i)
#pragma unroll
                for (int rep = 0; rep < 4; ++rep)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], b, a);
            for (int j = 0; j < 8; ++j) res += v[j];
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
}
template <int PK>
float run2(float* d, int iters, int which) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k2<PK>, dim3(256), dim3(512), 0, 0, d, 10, which);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k2<PK>, dim3(256), dim3(512), 0, 0, d, iters, which);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

// the same specialisation with a choice of matrix instruction: 0 f32 16x16x4, 1 f16 16x16x32, 2 f16 16x16x16, 3 bf16 16x16x32
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
template <int T>
__global__ __launch_bounds__(512) void k3(float* out, int iters, int which) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float a = threadIdx.x * 1e-3f, b = 1.0001f, res = 0.f;
    if ((wave & 4) == 0) {
        if (which & 1) {
            f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
            h8 x8, y8; h4 x4, y4; b8 p8, q8;
            for (int j = 0; j < 8; ++j) { x8[j] = (_Float16)(a + j); y8[j] = (_Float16)(b + j); p8[j] = (__bf16)(a + j); q8[j] = (__bf16)(b + j); }
            for (int j = 0; j < 4; ++j) { x4[j] = x8[j]; y4[j] = y8[j]; }
            for (int i = 0; i < iters; ++i) {
                if constexpr (T == 0) { c0 = MFMA(a, b, c0); c1 = MFMA(a, b, c1); c2 = MFMA(a, b, c2); c3 = MFMA(a, b, c3); }
                if constexpr (T == 1) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x8, y8, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x8, y8, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x8, y8, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x8, y8, c3, 0, 0, 0);
                }
                if constexpr (T == 2) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x16f16(x4, y4, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x16f16(x4, y4, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x16f16(x4, y4, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x16f16(x4, y4, c3, 0, 0, 0);
                }
                if constexpr (T == 3) {
                    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p8, q8, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p8, q8, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p8, q8, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p8, q8, c3, 0, 0, 0);
                }
            }
            res = c0[0] + c1[1] + c2[2] + c3[3];
        }
    } else if (which & 2) {
        float v[8];
        for (int j = 0; j < 8; ++j) v[j] = a + j;
        for (int i = 0; i < iters; ++i)
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], b, a);
        for (int j = 0; j < 8; ++j) res += v[j];
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
}
template <int T>
float run3(float* d, int iters, int which) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k3<T>, dim3(256), dim3(512), 0, 0, d, 10, which);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k3<T>, dim3(256), dim3(512), 0, 0, d, iters, which);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

template <int MODE>
float run(float* d, int iters, int block) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(block), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(block), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    const int iters = 20000;
    // 256 blocks = one per CU.  block 256 = 1 wave / SIMD, block 512 = 2 waves / SIMD (waves 0-3 and 4-7)
    printf("per wave: %d x (4 MFMA 16x16x4 f32 + 32 v_fma)\n", iters);
    printf("1 wave/SIMD   MFMA only   %.3f ms\n", run<0>(d, iters, 256));
    printf("1 wave/SIMD   VALU only   %.3f ms\n", run<1>(d, iters, 256));
    printf("1 wave/SIMD   interleaved %.3f ms\n", run<2>(d, iters, 256));
    printf("2 waves/SIMD  MFMA only   %.3f ms\n", run<0>(d, iters, 512));
    printf("2 waves/SIMD  VALU only   %.3f ms\n", run<1>(d, iters, 512));
    printf("2 waves/SIMD  interleaved %.3f ms\n", run<2>(d, iters, 512));
    printf("2 waves/SIMD  one MFMA wave + one VALU wave  %.3f ms\n", run<3>(d, iters, 512));
    printf("specialised waves (4 MFMA waves + 4 VALU waves per CU, one of each per SIMD), 32 v_fma per iteration:\n");
    printf("  MFMA waves only %.3f ms | VALU waves only %.3f ms | both %.3f ms\n", run2<0>(d, iters, 1), run2<0>(d, iters, 2), run2<0>(d, iters, 3));
    printf("same with 32 v_pk_fma_f32 (64 fma) per iteration:\n");
    printf("  MFMA waves only %.3f ms | VALU waves only %.3f ms | both %.3f ms\n", run2<1>(d, iters, 1), run2<1>(d, iters, 2), run2<1>(d, iters, 3));
    printf("specialised waves, 4 matrix instructions against 16 v_fma per iteration (MFMA waves | VALU waves | both):\n");
    printf("  f32 16x16x4    %.3f | %.3f | %.3f ms\n", run3<0>(d, iters, 1), run3<0>(d, iters, 2), run3<0>(d, iters, 3));
    printf("  f16 16x16x32   %.3f | %.3f | %.3f ms\n", run3<1>(d, iters, 1), run3<1>(d, iters, 2), run3<1>(d, iters, 3));
    printf("  f16 16x16x16   %.3f | %.3f | %.3f ms\n", run3<2>(d, iters, 1), run3<2>(d, iters, 2), run3<2>(d, iters, 3));
    printf("  bf16 16x16x32  %.3f | %.3f | %.3f ms\n", run3<3>(d, iters, 1), run3<3>(d, iters, 2), run3<3>(d, iters, 3));
    return 0;
}
