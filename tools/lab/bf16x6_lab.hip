// Lab: cost of accumulating one 32-entry group of B += V^T D V per wave, two ways (gfx950):
//   F32 : 8 k-steps of v_mfma_f32_16x16x4_f32 per tile, operands scaled by w on the VALU (what solve_directw does today)
//   BF6 : operands scaled by sqrt(w), split into three bf16 parts (hi, mid, lo) on the VALU, six
//         v_mfma_f32_16x16x32_bf16 per tile (hi.hi, hi.mid, mid.hi, hi.lo, mid.mid, lo.hi): fp32-equivalent accuracy
// hipcc --offload-arch=gfx950 -O3 -std=c++20 -o bf16x6_lab bf16x6_lab.hip && ./bf16x6_lab
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int NFB, int MODE, int OCC>
__global__ __launch_bounds__(64, OCC) void lab(const float* __restrict__ in, float* __restrict__ out, int iters) {
    constexpr int NT = NFB * (NFB + 1) / 2;
    const int lane = threadIdx.x;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float racc[NFB];
#pragma unroll
    for (int b = 0; b < NFB; ++b) racc[b] = 0.f;
    float x[8][NFB];                                   // 8 entries (k-steps for F32, consecutive k for BF6) x NFB feature blocks
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int b = 0; b < NFB; ++b) x[e][b] = in[(e * NFB + b) * 64 + lane];
    float w = in[lane] * 0.01f + 1.f, p = w + 1.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int b = 0; b < NFB; ++b) asm volatile("" : "+v"(x[e][b]));       // "freshly loaded": nothing may be hoisted
        if constexpr (MODE == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float fw[NFB];
#pragma unroll
                for (int b = 0; b < NFB; ++b) { fw[b] = x[e][b] * w; racc[b] += x[e][b] * p; }
                int t = 0;
#pragma unroll
                for (int bi = 0; bi < NFB; ++bi)
#pragma unroll
                    for (int bj = bi; bj < NFB; ++bj, ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[e][bi], fw[bj], acc[t], 0, 0, 0);
            }
        } else {
            bf16x8 hi[NFB], mid[NFB], lo[NFB];
#pragma unroll
            for (int b = 0; b < NFB; ++b) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    racc[b] += x[e][b] * p;
                    const float s = x[e][b] * w;                    // w stands for sqrt(w) here
                    const __bf16 h = (__bf16)s;
                    const float r1 = s - (float)h;
                    const __bf16 m = (__bf16)r1;
                    const float r2 = r1 - (float)m;
                    hi[b][e] = h; mid[b][e] = m; lo[b][e] = (__bf16)r2;
                }
            }
            int t = 0;
#pragma unroll
            for (int bi = 0; bi < NFB; ++bi)
#pragma unroll
                for (int bj = bi; bj < NFB; ++bj, ++t) {
                    f32x4 c = acc[t];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lo[bi], hi[bj], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mid[bi], mid[bj], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi[bi], lo[bj], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mid[bi], hi[bj], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi[bi], mid[bj], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hi[bi], hi[bj], c, 0, 0, 0);
                    acc[t] = c;
                }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
#pragma unroll
    for (int b = 0; b < NFB; ++b) s += racc[b];
    out[blockIdx.x * 64 + lane] = s;
}

template <int NFB, int MODE, int OCC>
void run(const float* in, float* out, const char* name) {
    const int iters = 2000, grid = 1024 * OCC;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((lab<NFB, MODE, OCC>), dim3(grid), dim3(64), 0, 0, in, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((lab<NFB, MODE, OCC>), dim3(grid), dim3(64), 0, 0, in, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: OCC waves x iters groups
    printf("NFB=%d %s OCC=%d: %.3f ms  -> %.0f ns per 32-entry group per SIMD-slot (%.0f cycles at 2.4 GHz)\n", NFB, name, OCC, ms,
           ms * 1e6 / (iters * OCC), ms * 1e6 / (iters * OCC) * 2.4);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 8 * 16 * 64 * 4); hipMalloc(&out, 1024 * 4 * 64 * 4);
    float h[8 * 16 * 64];
    for (int i = 0; i < 8 * 16 * 64; ++i) h[i] = 0.01f * ((i * 2654435761u >> 8) % 2001) / 1000.f - 0.01f;
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    run<4, 0, 1>(in, out, "F32"); run<4, 1, 1>(in, out, "BF6");
    run<4, 0, 3>(in, out, "F32"); run<4, 1, 3>(in, out, "BF6");
    run<8, 0, 1>(in, out, "F32"); run<8, 1, 1>(in, out, "BF6");
    run<9, 0, 1>(in, out, "F32"); run<9, 1, 1>(in, out, "BF6");
    return 0;
}
