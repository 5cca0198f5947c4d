// Probe: how exact is v_mfma_f32_16x16x32_f16 / _bf16?  (products of full-mantissa inputs, sums across k)
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/probe tools/lab/mfma_f16_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

// A[16][32], B[32][16] given as float arrays (values exactly representable); out D[16][16]
template <bool BF>
__global__ void k(const float* A, const float* B, const float* Cin, float* D) {
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    f32x4 c;
    for (int e = 0; e < 4; ++e) c[e] = Cin[(4 * q + e) * 16 + r];
    if constexpr (BF) {
        b8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (__bf16)A[r * 32 + 8 * q + j]; b[j] = (__bf16)B[(8 * q + j) * 16 + r]; }
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    } else {
        h8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = (_Float16)A[r * 32 + 8 * q + j]; b[j] = (_Float16)B[(8 * q + j) * 16 + r]; }
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
    for (int e = 0; e < 4; ++e) D[(4 * q + e) * 16 + r] = c[e];
}

static float rnd_h(int bits) {            // random value with `bits` significant bits in [1, 2)
    unsigned m = rand() & ((1u << (bits - 1)) - 1);
    return 1.0f + (float)m / (float)(1u << (bits - 1));
}

int main() {
    float *A, *B, *C, *D;
    hipMallocManaged(&A, 16 * 32 * 4); hipMallocManaged(&B, 32 * 16 * 4); hipMallocManaged(&C, 256 * 4); hipMallocManaged(&D, 256 * 4);
    for (int bf = 0; bf < 2; ++bf) {
        const int bits = bf ? 8 : 11;
        for (int test = 0; test < 4; ++test) {
            srand(1 + test);
            for (int i = 0; i < 512; ++i) { A[i] = 0; B[i] = 0; }
            for (int i = 0; i < 256; ++i) C[i] = 0;
            // test 0: one product per output (k = 0 only); 1: all 32 k, same magnitude; 2: 32 k with magnitudes 2^-(k%12);
            // 3: like 1 plus a large accumulator input (2^12)
            for (int i = 0; i < 16; ++i)
                for (int kk = 0; kk < 32; ++kk) {
                    if (test == 0 && kk > 0) continue;
                    float sa = (test == 2) ? ldexpf(1.f, -(kk % 12)) : 1.f;
                    A[i * 32 + kk] = rnd_h(bits) * sa * ((rand() & 1) ? 1.f : -1.f);
                    B[kk * 16 + i] = rnd_h(bits);
                }
            for (int i = 0; i < 16; ++i) for (int kk = 0; kk < 32; ++kk) if (!(test == 0 && kk > 0)) { for (int j = 0; j < 16; ++j) B[kk * 16 + j] = B[kk * 16 + (j % 16)] ; }
            for (int kk = 0; kk < 32; ++kk) for (int j = 0; j < 16; ++j) if (!(test == 0 && kk > 0)) B[kk * 16 + j] = rnd_h(bits);
            if (test == 3) for (int i = 0; i < 256; ++i) C[i] = 4096.f + (float)(rand() & 1023) / 1024.f;
            if (bf) hipLaunchKernelGGL(k<true>, dim3(1), dim3(64), 0, 0, A, B, C, D); else hipLaunchKernelGGL(k<false>, dim3(1), dim3(64), 0, 0, A, B, C, D);
            hipDeviceSynchronize();
            double worst = 0, worst_f32chain = 0;
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    double ex = C[i * 16 + j]; double mag = fabs(ex);
                    float chain = C[i * 16 + j];
                    for (int kk = 0; kk < 32; ++kk) { double p = (double)A[i * 32 + kk] * B[kk * 16 + j]; ex += p; mag += fabs(p); chain = fmaf(A[i * 32 + kk], B[kk * 16 + j], chain); }
                    double e = fabs(D[i * 16 + j] - ex) / mag;
                    if (e > worst) worst = e;
                    double e2 = fabs(chain - ex) / mag;
                    if (e2 > worst_f32chain) worst_f32chain = e2;
                }
            printf("%s test %d: worst |D - exact| / sum|terms| = %.3e (2^%.1f)   [f32 fma chain: %.3e]\n", bf ? "bf16" : "f16 ", test, worst, log2(worst + 1e-300), worst_f32chain);
        }
    }
    return 0;
}
